"""Wait-state rules around the inline-assembly memory instructions (no GPU: the device assembly is inspected).

The compiler keeps the hardware's wait states between its own instructions only.  k_inv_wave6 stores through an
inline-assembly `global_store_dwordx4`; the registers that held the data are free for the compiler to reuse at once, and
a VALU write to them inside the two wait states the store needs changes what is stored.  tools/check_asm_hazards.py
finds that pattern (and a VALU-written SGPR read too early by an inline-assembly memory instruction); here it runs on
the unit that contains the store, and on a snippet with the fault to show that it would be seen."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import check_asm_hazards as H  # noqa: E402

FAULTY = """
_Zkernel:
\ts_and_saveexec_b64 s[0:1], vcc
\t;;#ASMSTART
\tglobal_store_dwordx4 v252, v[2:5], s[80:81]
\t;;#ASMEND
.LBB0_1:
\ts_or_b64 exec, exec, s[0:1]
\tv_or_b32_e32 v2, 0x100, v251
\ts_endpgm
"""

FIXED = FAULTY.replace("\t;;#ASMEND", "\ts_nop 1\n\t;;#ASMEND")

SGPR_TOO_EARLY = """
_Zkernel:
\tv_readfirstlane_b32 s4, v0
\tv_readfirstlane_b32 s5, v1
\ts_nop 1
\t;;#ASMSTART
\tglobal_load_lds_dwordx4 v3, s[4:5]
\t;;#ASMEND
\ts_endpgm
"""


def _scan(tmp_path, text):
    p = tmp_path / "snippet.s"
    p.write_text(text)
    return H.scan(str(p))


def test_scanner_sees_the_store_data_hazard(tmp_path):
    assert _scan(tmp_path, FAULTY) == 1
    assert _scan(tmp_path, FIXED) == 0


def test_scanner_follows_branches(tmp_path):
    text = FAULTY.replace(".LBB0_1:\n", "\ts_cbranch_execnz .LBB0_2\n\ts_nop 3\n\ts_endpgm\n.LBB0_2:\n")
    assert _scan(tmp_path, text) == 0  # branch + s_or: two wait states on that path
    text = text.replace("\ts_or_b64 exec, exec, s[0:1]\n", "")
    assert _scan(tmp_path, text) == 1


def test_scanner_sees_a_valu_written_sgpr_read_too_early(tmp_path):
    assert _scan(tmp_path, SGPR_TOO_EARLY) == 2
    assert _scan(tmp_path, SGPR_TOO_EARLY.replace("s_nop 1", "s_nop 4")) == 0


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_wave_kernel_unit_has_no_inline_asm_hazard(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "vectorwave_amd", "csrc", "vw_launch_wave.hip")
    out = tmp_path / "vw_launch_wave.s"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-Wno-unused-function", "--cuda-device-only", "-S", "-o", str(out), src]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    text = out.read_text()
    assert "global_store_dwordx4" in text and ";;#ASMSTART" in text
    assert H.scan(str(out)) == 0
    # k_inv_wave6 counts its vector-memory operations by hand (s_waitcnt vmcnt(N) with N worked out from the LDS-DMA and
    # store instructions it issues): a register spilled to scratch would add loads and stores of the compiler's own to
    # that counter and make every count wrong.  No instantiation may use scratch.
    import re
    kernels = 0
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        if "k_inv_wave6" not in name:
            continue
        kernels += 1
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)) == 0, name
    assert kernels >= 12, kernels
