"""Machine check of the Java binding against the C header, without a JVM.

java/ai/prophetizo/wavelet/gpu/VwGpuNative.java declares one FFM downcall handle per exported function; a wrong
FunctionDescriptor (a missing parameter, JAVA_INT where the C side takes int64_t, ...) corrupts arguments
silently at run time.  This test parses every prototype of include/vwgpu.h and every
`h("symbol", FunctionDescriptor.of(...))` of the Java file and requires: every header symbol has exactly one handle,
no handle names a symbol the header lacks, and return type and each parameter agree in layout class
(pointer -> ADDRESS, int32_t/uint32_t -> JAVA_INT, int64_t/size_t -> JAVA_LONG, double -> JAVA_DOUBLE).  It also
checks that the facade classes only use handles that exist and that the status / flag constants equal the header's
enums."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vwgpu.h")
JAVA_DIR = os.path.join(ROOT, "java", "ai", "prophetizo", "wavelet")
NATIVE = os.path.join(JAVA_DIR, "gpu", "VwGpuNative.java")


def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"^\s*#.*$", " ", text, flags=re.M)


def _layout_of_ctype(t):
    t = " ".join(t.replace("const", " ").split())
    if "*" in t:
        return "ADDRESS"
    table = {"int32_t": "JAVA_INT", "uint32_t": "JAVA_INT", "int": "JAVA_INT", "int64_t": "JAVA_LONG",
             "size_t": "JAVA_LONG", "double": "JAVA_DOUBLE", "void": "VOID"}
    assert t in table, "C type %r has no FFM layout in this check" % t
    return table[t]


def header_prototypes():
    """{symbol: [return layout, parameter layouts...]}"""
    src = _strip_c(open(HEADER).read())
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \t\n\*]*?)\b(vw_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        lay = [_layout_of_ctype(ret)]
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                pm = re.match(r"(.*?)(\w+)$", a)  # type, then the parameter name
                assert pm and pm.group(1).strip(), "cannot split parameter %r of %s" % (a, name)
                lay.append(_layout_of_ctype(pm.group(1)))
        assert name not in out, "duplicate prototype " + name
        out[name] = lay
    return out


def java_descriptors():
    """{symbol: (constant name, [return layout, parameter layouts...])}"""
    src = re.sub(r"//[^\n]*", " ", open(NATIVE).read())
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    pat = r"static\s+final\s+MethodHandle\s+(\w+)\s*=\s*h\(\s*\"(\w+)\"\s*,\s*FunctionDescriptor\.(ofVoid|of)\(([^;]*?)\)\s*\)\s*;"
    for m in re.finditer(pat, src):
        const, sym, kind, body = m.groups()
        lay = [x.strip() for x in body.split(",") if x.strip()]
        if kind == "ofVoid":
            lay = ["VOID"] + lay
        for x in lay:
            assert x in ("ADDRESS", "JAVA_INT", "JAVA_LONG", "JAVA_DOUBLE", "VOID"), "%s: unexpected layout %s" % (sym, x)
        assert sym not in out, "two handles for " + sym
        out[sym] = (const, lay)
    return out


def test_every_header_symbol_has_a_matching_descriptor():
    protos = header_prototypes()
    descs = java_descriptors()
    assert len(protos) >= 45  # the parser found the header's functions, not a fragment
    missing = sorted(set(protos) - set(descs))
    extra = sorted(set(descs) - set(protos))
    assert not missing, "header symbols without a Java handle: %s" % missing
    assert not extra, "Java handles for symbols the header does not declare: %s" % extra
    for sym, lay in protos.items():
        const, jlay = descs[sym]
        assert len(jlay) == len(lay), "%s: %d layouts in Java, %d in the header" % (sym, len(jlay), len(lay))
        for i, (j, c) in enumerate(zip(jlay, lay)):
            where = "return value" if i == 0 else "parameter %d" % i
            assert j == c, "%s %s: Java %s, header %s" % (sym, where, j, c)
        assert const == sym.upper()


def test_library_exports_what_the_handles_bind():
    """The shared library is built in this tree: every bound symbol resolves (no compute is called)."""
    import ctypes as C
    lib_path = os.path.join(ROOT, "vectorwave_amd", "libvwgpu.so")
    if not os.path.exists(lib_path):
        import pytest
        pytest.skip("libvwgpu.so not built")
    lib = C.CDLL(lib_path)
    for sym in java_descriptors():
        assert hasattr(lib, sym), sym


def test_java_sources_use_only_declared_handles_and_constants():
    descs = java_descriptors()
    consts = {c for c, _ in descs.values()}
    native_src = open(NATIVE).read()
    declared_fields = set(re.findall(r"\b([A-Z][A-Z0-9_]+)\s*=", native_src))
    declared_methods = set(re.findall(r"static\s+[\w<>\[\]]+\s+(\w+)\s*\(", native_src))
    seen_handles = set()
    for dirpath, _, files in os.walk(JAVA_DIR):
        for f in files:
            if not f.endswith(".java") or f == "VwGpuNative.java":
                continue
            src = open(os.path.join(dirpath, f)).read()
            for name in re.findall(r"VwGpuNative\.(\w+)", src):
                if name.isupper() or "_" in name:
                    assert name in consts or name in declared_fields, "%s uses VwGpuNative.%s, which is not declared" % (f, name)
                    if name in consts:
                        seen_handles.add(name)
                else:
                    assert name in declared_methods, "%s calls VwGpuNative.%s(), which is not declared" % (f, name)
    # the drop-in classes and the backend reach the hot path
    for needed in ("VW_MODWT_FORWARD_BATCH", "VW_MODWT_INVERSE_BATCH", "VW_MODWT_INVERSE_BANDS_BATCH", "VW_SWT_DENOISE_BATCH",
                   "VW_SWT_UNIVERSAL_THRESHOLD_BATCH", "VW_MAX_LEVELS", "VW_DESTROY", "VW_WAVELET_DENOISE_BATCH",
                   "VW_STREAMING_DENOISE_BATCH", "VW_STREAM_STATE_CREATE", "VW_STREAM_STATE_DESTROY", "VW_BATCH_MODWT_SOA"):
        assert needed in seen_handles, needed + " is bound but never called from the Java host side"


def test_invoke_sites_pass_as_many_arguments_as_the_descriptor_has():
    """Every `VwGpuNative.X.invokeExact(args)` in the Java host code passes exactly the descriptor's parameter count."""
    descs = {c: lay for c, lay in java_descriptors().values()}
    checked = 0
    for dirpath, _, files in os.walk(JAVA_DIR):
        for f in files:
            if not f.endswith(".java"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            for m in re.finditer(r"(?:VwGpuNative\.)?\b(VW_[A-Z0-9_]+)\.invokeExact\(", src):
                const = m.group(1)
                assert const in descs, "%s invokes %s, which is not a handle" % (f, const)
                i, depth, nargs, cur = m.end(), 1, 0, ""
                while depth:
                    ch = src[i]
                    if ch in "([{":
                        depth += 1
                    elif ch in ")]}":
                        depth -= 1
                        if depth == 0:
                            break
                    if ch == "," and depth == 1:
                        nargs += 1
                        cur = ""
                    else:
                        cur += ch
                    i += 1
                if cur.strip():
                    nargs += 1
                assert nargs == len(descs[const]) - 1, "%s: %s.invokeExact passes %d arguments, the descriptor has %d" % (
                    f, const, nargs, len(descs[const]) - 1)
                checked += 1
    assert checked >= 15


def test_constants_equal_the_header_enums():
    hdr = _strip_c(open(HEADER).read())
    enums = dict((k, int(v)) for k, v in re.findall(r"\b(VW_[A-Z_]+)\s*=\s*(\d+)", hdr))
    src = open(NATIVE).read()
    java = dict((k, int(v)) for k, v in re.findall(r"\b([A-Z][A-Z_]+)\s*=\s*(\d+)", src))
    for name, value in enums.items():
        if name.startswith("VW_E_") or name == "VW_OK":
            assert java.get(name) == value, name
        elif name.startswith("VW_FLAG_"):
            assert java.get(name[3:]) == value, name
        elif name.startswith("VW_BOUNDARY_"):
            assert java.get(name[3:]) == value, name
        elif name.startswith("VW_NOISE_"):
            assert java.get(name[3:]) == value, name


def test_generated_block_is_current():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_java_handles.py"), "--check"])
    assert r.returncode == 0, "VwGpuNative.java's handle block is stale: run tools/gen_java_handles.py"


def test_facade_classes_keep_the_reference_signatures():
    """The public surface of the drop-in classes, as the reference declares it (signatures listed from
    modwt/MODWTTransform.java:83-504, modwt/MultiLevelMODWTTransform.java:186-785, swt/VectorWaveSwtAdapter.java:122-423,
    denoising/WaveletDenoiser.java:79-634, modwt/streaming/MODWTStreamingDenoiser.java:35-506,
    modwt/BatchSIMDMODWT.java:64-328, modwt/ParallelMultiLevelMODWT.java:37-60)."""
    want = {
        "modwt/MODWTTransform.java": [
            "public MODWTTransform(Wavelet wavelet, BoundaryMode boundaryMode)",
            "public MODWTResult forward(double[] signal)",
            "public double[] inverse(MODWTResult modwtResult)",
            "public Wavelet getWavelet()",
            "public BoundaryMode getBoundaryMode()",
            "public WaveletOperations.PerformanceInfo getPerformanceInfo()",
            "public ProcessingEstimate estimateProcessingTime(int signalLength)",
            "public MODWTResult[] forwardBatch(double[][] signals)",
            "public double[][] inverseBatch(MODWTResult[] results)",
        ],
        "modwt/MultiLevelMODWTTransform.java": [
            "public MultiLevelMODWTTransform(Wavelet wavelet, BoundaryMode boundaryMode)",
            "public MultiLevelMODWTResult decompose(double[] signal)",
            "public MultiLevelMODWTResult decompose(double[] signal, int levels)",
            "public MutableMultiLevelMODWTResult decomposeMutable(double[] signal)",
            "public MutableMultiLevelMODWTResult decomposeMutable(double[] signal, int levels)",
            "public double[] reconstruct(MultiLevelMODWTResult result)",
            "public double[] reconstructFromLevel(MultiLevelMODWTResult result, int startLevel)",
            "public double[] reconstructLevels(MultiLevelMODWTResult result, int minLevel, int maxLevel)",
            "public Wavelet getWavelet()",
            "public BoundaryMode getBoundaryMode()",
            "public int getMaximumLevels(int signalLength)",
            "public static int getMaxDecompositionLevels()",
        ],
        "swt/VectorWaveSwtAdapter.java": [
            "public VectorWaveSwtAdapter(Wavelet wavelet, BoundaryMode boundaryMode)",
            "public VectorWaveSwtAdapter(Wavelet wavelet)",
            "public MutableMultiLevelMODWTResult forward(double[] signal)",
            "public MutableMultiLevelMODWTResult forward(double[] signal, int levels)",
            "public double[] inverse(MutableMultiLevelMODWTResult result)",
            "public void applyThreshold(MutableMultiLevelMODWTResult result, int level, double threshold, boolean soft)",
            "public void applyUniversalThreshold(MutableMultiLevelMODWTResult result, boolean soft)",
            "public double[] denoise(double[] signal, int levels)",
            "public double[] denoise(double[] signal, int levels, double threshold, boolean soft)",
            "public double[] extractLevel(double[] signal, int levels, int targetLevel)",
            "public Wavelet getWavelet()",
            "public BoundaryMode getBoundaryMode()",
            "public void cleanup()",
            "public Map<String, Object> getCacheStatistics()",
        ],
        "denoising/WaveletDenoiser.java": [
            "public class WaveletDenoiser",
            "public WaveletDenoiser(Wavelet wavelet, BoundaryMode boundaryMode)",
            "public static WaveletDenoiser forFinancialData()",
            "public double[] denoise(double[] signal, ThresholdMethod method)",
            "public double[] denoise(double[] signal, ThresholdMethod method, ThresholdType type)",
            "public double[] denoiseMultiLevel(double[] signal, int levels, ThresholdMethod method, ThresholdType type)",
            "public double[] denoiseFixed(double[] signal, double threshold, ThresholdType type)",
            "public enum ThresholdMethod",
            "public enum ThresholdType",
        ],
        "modwt/streaming/MODWTStreamingDenoiser.java": [
            "public class MODWTStreamingDenoiser implements Flow.Publisher<double[]>, AutoCloseable",
            "public enum NoiseEstimation",
            "private MODWTStreamingDenoiser(Builder builder)",
            "public double[] denoise(double[] samples)",
            "public double getEstimatedNoiseLevel()",
            "public long getSamplesProcessed()",
            "public void subscribe(Flow.Subscriber<? super double[]> subscriber)",
            "public void close()",
            "public boolean isClosed()",
            "public static class Builder",
            "public Builder wavelet(Wavelet wavelet)",
            "public Builder boundaryMode(BoundaryMode boundaryMode)",
            "public Builder bufferSize(int bufferSize)",
            "public Builder thresholdType(ThresholdType thresholdType)",
            "public Builder thresholdMethod(ThresholdMethod thresholdMethod)",
            "public Builder thresholdMultiplier(double thresholdMultiplier)",
            "public Builder noiseEstimation(NoiseEstimation noiseEstimation)",
            "public Builder noiseWindowSize(int noiseWindowSize)",
            "public MODWTStreamingDenoiser build()",
        ],
        "modwt/BatchSIMDMODWT.java": [
            "public final class BatchSIMDMODWT",
            "public static void batchMODWTSoA(double[] soaSignals, double[] soaApprox, double[] soaDetail, "
            "DiscreteWavelet wavelet, int batchSize, int signalLength)",
            "public static void convertToSoA(double[][] signals, double[] soaOutput)",
            "public static void convertFromSoA(double[] soaData, double[][] output)",
            "public static void batchMODWTWithCleanup(double[] soaSignals, double[] soaApprox, double[] soaDetail, "
            "DiscreteWavelet wavelet, int batchSize, int signalLength)",
        ],
        "modwt/ParallelMultiLevelMODWT.java": [
            "public class ParallelMultiLevelMODWT",
            "public ParallelMultiLevelMODWT()",
            "public ParallelMultiLevelMODWT(Executor executor)",
            "public MultiLevelMODWTResult decompose(double[] signal, Wavelet wavelet, BoundaryMode mode, int levels)",
        ],
    }
    # enum constants in the reference's order: their ordinals cross the C ABI as VW_THRESHOLD_* / VW_NOISE_*
    den = " ".join(re.sub(r"/\*.*?\*/", " ", open(os.path.join(JAVA_DIR, "denoising/WaveletDenoiser.java")).read(), flags=re.S).split())
    assert re.search(r"enum ThresholdMethod \{ UNIVERSAL, SURE, MINIMAX, BAYES, FIXED \}", den), "ThresholdMethod order"
    assert re.search(r"enum ThresholdType \{ SOFT, HARD \}", den), "ThresholdType order"
    stream = " ".join(re.sub(r"/\*.*?\*/", " ", open(os.path.join(JAVA_DIR, "modwt/streaming/MODWTStreamingDenoiser.java")).read(),
                             flags=re.S).split())
    assert re.search(r"enum NoiseEstimation \{ MAD, STD, FIXED \}", stream), "NoiseEstimation order"
    hdr = _strip_c(open(HEADER).read())
    enums = dict((k, int(v)) for k, v in re.findall(r"\b(VW_[A-Z_]+)\s*=\s*(\d+)", hdr))
    for i, name in enumerate(["UNIVERSAL", "SURE", "MINIMAX", "BAYES", "FIXED"]):
        assert enums["VW_THRESHOLD_" + name] == i
    for i, name in enumerate(["MAD", "STD", "FIXED"]):
        assert enums["VW_NOISE_" + name] == i
    for rel, sigs in want.items():
        src = " ".join(open(os.path.join(JAVA_DIR, rel)).read().split())
        for s in sigs:
            assert s in src, "%s lacks `%s`" % (rel, s)


def _split_top_level(argtext):
    depth, cur, out = 0, "", []
    for ch in argtext:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _call_args(src, start):
    """Text between the parenthesis at src[start - 1] and its partner."""
    i, depth = start, 1
    while depth:
        ch = src[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        i += 1
    return src[start:i - 1]


def test_facades_call_backend_methods_that_exist_with_that_many_arguments():
    """No JDK here, so the cheapest compile error is checked by hand: every call of a GpuModwtBackend method from the
    drop-in classes names a public method of the backend and passes as many arguments as one of its overloads takes."""
    backend_src = open(os.path.join(JAVA_DIR, "gpu", "GpuModwtBackend.java")).read()
    backend_src = re.sub(r"/\*.*?\*/", " ", backend_src, flags=re.S)
    backend_src = re.sub(r"//[^\n]*", " ", backend_src)
    declared = {}
    for m in re.finditer(r"public\s+(?:static\s+)?[\w<>\[\], ]+?\s+(\w+)\s*\(", backend_src):
        args = _call_args(backend_src, m.end())
        declared.setdefault(m.group(1), set()).add(len(_split_top_level(args)))
    assert {"forward", "inverse", "denoise", "waveletDenoise", "streamingDenoise", "batchMODWTSoA"} <= set(declared)
    calls = 0
    for dirpath, _, files in os.walk(JAVA_DIR):
        for f in files:
            if not f.endswith(".java") or f in ("GpuModwtBackend.java", "VwGpuNative.java"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
            src = re.sub(r"//[^\n]*", " ", src)
            for m in re.finditer(r"(?:GpuModwtBackend\s*\.\s*shared\(\)|\bbackend)\s*\.\s*(\w+)\s*\(", src):
                name = m.group(1)
                nargs = len(_split_top_level(_call_args(src, m.end())))
                assert name in declared, "%s calls GpuModwtBackend.%s, which the backend does not declare" % (f, name)
                assert nargs in declared[name], "%s: %s called with %d arguments, declared with %s" % (
                    f, name, nargs, sorted(declared[name]))
                calls += 1
    assert calls >= 20
