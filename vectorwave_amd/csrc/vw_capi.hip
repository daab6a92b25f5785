// vw_capi.hip -- C ABI (include/vwgpu.h) over the gfx950 kernels.
//
// Host side of the engine: argument validation with the reference's error
// codes, 1/sqrt(2) tap scaling, level grouping + halo planning for the fused
// tile kernels, kernel-instance dispatch, staging for the host-pointer entry
// points.  No CPU compute path exists here: without a HIP device every entry
// point that would compute fails with VW_E_NO_DEVICE.
#include "vw_aux_kernels.hip.h"
#include "vw_launch.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/vwgpu.h"

using namespace vw;

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Choices among equivalent launch plans (tile shape, run lengths, kernel family).  Every setting produces the
// same bits; the defaults are the measured-best ones.  Set through vw_set_plan_option (tests exercise the
// rarely-taken plans on small inputs with it); the product library never reads the environment.
struct PlanOptions {
    int fwd_tile_p = 0, fwd_tile_h = 0;  // forced forward tile shape (chunks per tile, halo chunks); 0 = planner's choice
    int inv_tile_p = 0, inv_tile_h = 0;
    int64_t fwd_chunks = 0;              // cap on the tiles per run of the streaming forward; 0 = default
    int64_t stream_wgs = 0;              // workgroups wanted per streaming launch; 0 = default
    int64_t sdry_chunks = 0;             // run length of the long-context kernels; 0 = default
    int no_sdry = 0;                     // 1: never use the long-context kernels
    int nt_stores = -1;                  // -1 by launch size, 0/1 forced
    int no_unroll = 0;                   // 1: generic tap loop also for 8+8-tap filters
    int no_reg_select = 0;               // 1: LDS/global selection kernels also for short rows
    int select_lds_cap = 0;              // longest row the 256-lane selection kernel copies into LDS; 0 = default
    int no_wg_select = 0;                // 1: rows of 2049..32768 samples through the 256-lane radix select, not the workgroup-resident descent
    int no_std_chains = 0;               // 1: plain reduction kernel for the streaming STD
    int no_row_denoise = 0;              // 1: the SWT denoise always materialises the coefficient block
    int no_blk = 0;                      // 1: the inverse tile kernel without register blocking (k_inv_fused)
    int no_ring = 0;                     // 1: bands land in registers (k_inv_blk) instead of the LDS-DMA ring
    int inv_wave = 0;                    // one wavefront per run of 512-sample chunks (k_inv_wave6): 0 = where it pays (large
                                         // batches), -1 = never, N >= 1 = wherever it applies, runs of N chunks
    int inv_wave_nt = -1;                // k_inv_wave6 loads its bands with the streaming hint: -1 = as the stores (large launches), 0 / 1
    int carry_chunks = 0;                // >= 2: the ring kernel over runs of that many tiles with carried right contexts
                                         // (k_inv_carry); measured 5-7 % slower than one tile per workgroup, so off by default
    int no_wide_inv = 0;                 // 1: long-context inverse never uses the 1024-lane shape
    int inv_vm = 0;                      // global access mode of the long-context inverse: 0 one sample per access
                                         // (measured 6 % faster than pairs, aligned or not), 1 / 2 pairs, -1 by alignment
    int no_band_regs = 0;                // 1: the row-resident denoise keeps every detail band in LDS
    int no_vec2 = 0;                     // 1: the inverse tile kernels access global memory one sample at a time
    int no_sym_tiles = 0;                // 1: SYMMETRIC always through the generic one-level kernels
    int no_soa_runs = 0;                 // 1: the structure-of-arrays transform one output per lane (k_soa_modwt)
    int soa_run_len = 0;                 // samples per lane of k_soa_modwt_run: 8, 16 or 32; 0 = default
#ifdef VW_TUNING
    char ablate[64] = {0};               // tuning build only: switches parts of a kernel off (wrong results)
#endif
};

struct vw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    char info[512] = {0};
    char plan_fwd[96] = {0};             // kernels the last forward / inverse call enqueued (vw_last_plan)
    char plan_inv[96] = {0};
    char plan_txt[224] = {0};
    std::recursive_mutex mu;  // serialises use of the context-owned buffers below
    DevBuf ws;          // ping-pong plane [B][N] for multi-group / generic plans
    DevBuf stage_in;    // host-pointer entry points: device copy of the input
    DevBuf stage_out;   // ... and of the output
    DevBuf denoise_ws;  // coefficient block for vw_swt_denoise_batch(_dev) when the caller passes none
    DevBuf cell;        // 8-byte first_bad cell for host-pointer entry points
    DevBuf ws_borrow;   // set (under mu) by an entry point that may not allocate: stands in for `ws`, never freed
    void* scratch = nullptr;  // this device's copy of the static scratch array (looked up in vw_init)
    int lds_limit = 65536;
    int num_cus = 256;
    PlanOptions opt;
};

static int set_err(vw_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx != nullptr) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define VW_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return set_err((ctx), e_ == hipErrorOutOfMemory ? VW_E_NOMEM : VW_E_HIP, "%s: %s", #call, \
                           hipGetErrorString(e_));                                               \
    } while (0)

static int ensure(vw_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p != nullptr) return VW_OK;
    if (b.p != nullptr) {
        VW_HIP(ctx, hipDeviceSynchronize());  // the buffer may be in use on a caller's stream
        VW_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0) bytes = 8;
    VW_HIP(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return VW_OK;
}

// The context's ping-pong plane, or the stand-in an allocation-free entry point has lent for this call.
static int ensure_ws(vw_ctx* ctx, size_t bytes, double** out) {
    if (ctx->ws_borrow.p != nullptr) {
        if (ctx->ws_borrow.bytes < bytes)
            return set_err(ctx, VW_E_WORKSPACE, "this plan needs a %zu-byte intermediate plane, %zu bytes of static scratch are left for it",
                           bytes, ctx->ws_borrow.bytes);
        *out = static_cast<double*>(ctx->ws_borrow.p);
        return VW_OK;
    }
    const int rc = ensure(ctx, ctx->ws, bytes);
    *out = static_cast<double*>(ctx->ws.p);
    return rc;
}

// ---------------------------------------------------------------------------
// static device scratch of the allocation-free entry point
// ---------------------------------------------------------------------------
//
// One array per GPU, part of this code object, held from the moment the library is loaded on that GPU until
// the process ends.  `vw_scratch` is only a symbol on the host side: its device address differs per GPU and is
// looked up once per device in vw_init (that device current) and kept in g_scratch_ptr[]; every use below goes
// through that pointer.
static constexpr size_t kScratchBytes = VW_NOALLOC_SCRATCH_BYTES;
static constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char vw_scratch[kScratchBytes];
static void* g_scratch_ptr[kMaxDevices] = {nullptr};
static std::mutex g_scratch_mu;

static int scratch_lookup(int device, void** out) {
    if (device < 0 || device >= kMaxDevices) return VW_E_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device] == nullptr) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(vw_scratch));  // the current device's copy
        if (e != hipSuccess || p == nullptr) {
            (void)hipGetLastError();
            return VW_E_HIP;
        }
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return VW_OK;
}

static bool parse_pair(int64_t v, int* p, int* h) {
    // tile shapes travel as ppt * 100 + hpt (e.g. 401 = 4 chunks per tile, 1 halo chunk); 0 clears
    if (v < 0 || v > 99999) return false;
    *p = static_cast<int>(v / 100);
    *h = static_cast<int>(v % 100);
    return true;
}

extern "C" int32_t vw_set_plan_option(vw_ctx* ctx, const char* name, int64_t value) {
    if (ctx == nullptr) return VW_E_NULL;
    if (name == nullptr) return set_err(ctx, VW_E_NULL, "option name cannot be null");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    PlanOptions& o = ctx->opt;
    if (strcmp(name, "reset") == 0) {
        o = PlanOptions();
#ifdef VW_TUNING
        if (const char* e = getenv("VW_ABLATE")) snprintf(o.ablate, sizeof(o.ablate), "%s", e);  // ablations survive a reset
#endif
    } else if (strcmp(name, "fwd_tile") == 0) {
        if (!parse_pair(value, &o.fwd_tile_p, &o.fwd_tile_h)) return set_err(ctx, VW_E_UNSUPPORTED, "bad tile shape %lld", static_cast<long long>(value));
    } else if (strcmp(name, "inv_tile") == 0) {
        if (!parse_pair(value, &o.inv_tile_p, &o.inv_tile_h)) return set_err(ctx, VW_E_UNSUPPORTED, "bad tile shape %lld", static_cast<long long>(value));
    } else if (strcmp(name, "fwd_chunks") == 0) {
        o.fwd_chunks = value;
    } else if (strcmp(name, "stream_wgs") == 0) {
        o.stream_wgs = value;
    } else if (strcmp(name, "sdry_chunks") == 0) {
        o.sdry_chunks = value;
    } else if (strcmp(name, "no_sdry") == 0) {
        o.no_sdry = value != 0;
    } else if (strcmp(name, "nt_stores") == 0) {
#ifdef VW_TUNING
        o.nt_stores = value < 0 ? -1 : static_cast<int>(value & 7);  // tuning build: k_inv_wave6 knows other policy bits
#else
        o.nt_stores = value < 0 ? -1 : (value != 0);
#endif
    } else if (strcmp(name, "no_unroll") == 0) {
        o.no_unroll = value != 0;
    } else if (strcmp(name, "no_reg_select") == 0) {
        o.no_reg_select = value != 0;
    } else if (strcmp(name, "no_wg_select") == 0) {
        o.no_wg_select = value != 0;
    } else if (strcmp(name, "select_lds_cap") == 0) {
        o.select_lds_cap = value > 0 ? static_cast<int>(value) : 0;
    } else if (strcmp(name, "no_std_chains") == 0) {
        o.no_std_chains = value != 0;
    } else if (strcmp(name, "no_row_denoise") == 0) {
        o.no_row_denoise = value != 0;
    } else if (strcmp(name, "no_blk") == 0) {
        o.no_blk = value != 0;
    } else if (strcmp(name, "no_ring") == 0) {
        o.no_ring = value != 0;
    } else if (strcmp(name, "inv_wave") == 0) {
        o.inv_wave = value > 0 ? static_cast<int>(value < 1000000 ? value : 1000000) : (value < 0 ? -1 : 0);
    } else if (strcmp(name, "inv_wave_nt") == 0) {
#ifdef VW_TUNING
        o.inv_wave_nt = value < 0 ? -1 : static_cast<int>(value & 7);  // tuning build: any combination of the policy bits
#else
        o.inv_wave_nt = value < 0 ? -1 : (value != 0);
#endif
    } else if (strcmp(name, "carry_chunks") == 0) {
        o.carry_chunks = value > 0 ? static_cast<int>(value) : 0;
    } else if (strcmp(name, "no_wide_inv") == 0) {
        o.no_wide_inv = value != 0;
    } else if (strcmp(name, "inv_vm") == 0) {
        o.inv_vm = (value >= 0 && value <= 2) ? static_cast<int>(value) : -1;
    } else if (strcmp(name, "no_vec2") == 0) {
        o.no_vec2 = value != 0;
    } else if (strcmp(name, "no_band_regs") == 0) {
        o.no_band_regs = value != 0;
    } else if (strcmp(name, "no_sym_tiles") == 0) {
        o.no_sym_tiles = value != 0;
    } else if (strcmp(name, "no_soa_runs") == 0) {
        o.no_soa_runs = value != 0;
    } else if (strcmp(name, "soa_run_len") == 0) {
        o.soa_run_len = (value == 8 || value == 16 || value == 32) ? static_cast<int>(value) : 0;
    } else {
        return set_err(ctx, VW_E_UNSUPPORTED, "unknown plan option '%s'", name);
    }
    return VW_OK;
}

#ifdef VW_TUNING
// Tuning build (tools/ only, never shipped as libvwgpu.so): the plan options can also come from the environment,
// and VW_ABLATE switches parts of the kernels off to time what is left -- with wrong results.
static void tuning_env(vw_ctx* ctx) {
    static const char* const names[] = {"fwd_chunks", "stream_wgs", "sdry_chunks", "no_sdry", "nt_stores", "no_unroll",
                                        "no_reg_select", "no_wg_select", "select_lds_cap", "no_std_chains", "no_row_denoise", "no_blk", "no_ring", "inv_wave", "inv_wave_nt", "carry_chunks", "no_wide_inv", "inv_vm", "no_vec2", "no_band_regs", "no_sym_tiles", "no_soa_runs", "soa_run_len"};
    for (const char* nm : names) {
        char env[64] = "VW_";
        size_t k = 3;
        for (const char* c = nm; *c != 0 && k + 1 < sizeof(env); ++c) env[k++] = static_cast<char>(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
        env[k] = 0;
        if (const char* e = getenv(env)) (void)vw_set_plan_option(ctx, nm, atoll(e));
    }
    for (int inv = 0; inv < 2; ++inv)
        if (const char* e = getenv(inv ? "VW_INV_TILE" : "VW_FWD_TILE")) {
            int p = 0, h = 0;
            if (sscanf(e, "%d,%d", &p, &h) == 2) (void)vw_set_plan_option(ctx, inv ? "inv_tile" : "fwd_tile", p * 100 + h);
        }
    if (const char* e = getenv("VW_ABLATE")) snprintf(ctx->opt.ablate, sizeof(ctx->opt.ablate), "%s", e);
}
#endif

extern "C" int32_t vw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int32_t vw_init(int32_t device, vw_ctx** out_ctx) {
    if (out_ctx == nullptr) return VW_E_NULL;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return VW_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return VW_E_NO_DEVICE;
    vw_ctx* ctx = new (std::nothrow) vw_ctx();
    if (ctx == nullptr) return VW_E_NOMEM;
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return VW_E_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->lds_limit = static_cast<int>(prop.sharedMemPerBlock);
        if (prop.multiProcessorCount > 0) ctx->num_cus = prop.multiProcessorCount;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0 && ctx->lds_limit < 160 * 1024) ctx->lds_limit = 160 * 1024;
        snprintf(ctx->info, sizeof(ctx->info),
                 "HIP device %d: %s (%s), %d CUs, %.0f GiB HBM, %d KiB LDS/workgroup; fp64 MODWT kernels: "
                 "LDS-tiled fused levels, no FMA contraction",
                 device, prop.name, prop.gcnArchName, prop.multiProcessorCount,
                 static_cast<double>(prop.totalGlobalMem) / (1024.0 * 1024.0 * 1024.0), ctx->lds_limit / 1024);
    } else {
        snprintf(ctx->info, sizeof(ctx->info), "HIP device %d", device);
    }
    strcpy(ctx->err, "no error");
    // the static array behind the allocation-free entry point.  A failed lookup is not fatal: every other entry point
    // works without it, and vw_swt_denoise_batch_noalloc_dev then reports VW_E_WORKSPACE for the batches that need it
    void* scratch = nullptr;
    if (scratch_lookup(device, &scratch) != VW_OK) scratch = nullptr;  // `device` is current here
    ctx->scratch = scratch;
#ifdef VW_TUNING
    tuning_env(ctx);
    strncat(ctx->info, " [TUNING BUILD: ablation hooks enabled]", sizeof(ctx->info) - strlen(ctx->info) - 1);
#endif
    *out_ctx = ctx;
    return VW_OK;
}

extern "C" int32_t vw_destroy(vw_ctx* ctx) {
    if (ctx == nullptr) return VW_E_NULL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->ws, &ctx->stage_in, &ctx->stage_out, &ctx->denoise_ws, &ctx->cell};
    for (DevBuf* b : bufs)
        if (b->p != nullptr) (void)hipFree(b->p);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VW_OK;
}

extern "C" const char* vw_last_error(const vw_ctx* ctx) { return ctx != nullptr ? ctx->err : "null context"; }
extern "C" const char* vw_performance_info(const vw_ctx* ctx) { return ctx != nullptr ? ctx->info : "null context"; }

// appends "name[lev0-lev1]" to one of the context's plan notes
static void plan_note(char (&dst)[96], const char* name, int lev0, int lev1) {
    const size_t used = strlen(dst);
    if (used + 1 >= sizeof(dst)) return;
    snprintf(dst + used, sizeof(dst) - used, "%s%s[%d-%d]", used ? "+" : "", name, lev0, lev1);
}

extern "C" const char* vw_last_plan(vw_ctx* ctx) {
    if (ctx == nullptr) return "null context";
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    snprintf(ctx->plan_txt, sizeof(ctx->plan_txt), "forward: %s; inverse: %s", ctx->plan_fwd[0] ? ctx->plan_fwd : "-",
             ctx->plan_inv[0] ? ctx->plan_inv : "-");
    return ctx->plan_txt;
}

extern "C" int32_t vw_synchronize(vw_ctx* ctx) {
    if (ctx == nullptr) return VW_E_NULL;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

// MultiLevelMODWTTransform.calculateMaxLevels (modwt/MultiLevelMODWTTransform.java:477-523)
extern "C" int32_t vw_host_alloc(vw_ctx* ctx, size_t bytes, void** out) {
    if (ctx == nullptr) return VW_E_NULL;
    if (out == nullptr) return set_err(ctx, VW_E_NULL, "result pointer cannot be null");
    *out = nullptr;
    if (bytes == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return set_err(ctx, VW_E_NOMEM, "cannot page-lock %zu bytes of host memory", bytes);
    }
    return VW_OK;
}

extern "C" int32_t vw_host_free(vw_ctx* ctx, void* ptr) {
    if (ctx == nullptr) return VW_E_NULL;
    if (ptr == nullptr) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    VW_HIP(ctx, hipHostFree(ptr));
    return VW_OK;
}

extern "C" int32_t vw_max_levels(int64_t n, int32_t filter_len) {
    if (n <= filter_len) return 0;
    int max_level = 1;
    const int64_t lm1 = filter_len - 1;
    while (max_level < 10) {
        if (max_level - 1 >= 31) break;
        const int64_t scaled = lm1 * (static_cast<int64_t>(1) << (max_level - 1)) + 1;
        if (scaled > n) break;
        ++max_level;
    }
    return max_level - 1;
}

// ---------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------

static constexpr int kHaloMaxChunks = 10;  // largest HPT instantiated

// Instantiated shapes, each list in order of preference within one halo size (measured on MI355X, DB4
// J=6, 65536 x 16384).  Keep in sync with the dispatch below.
//   forward: 1024-sample tiles streamed along the row beat larger ones (more resident workgroups)
//   inverse, reference arithmetic: 2048-sample tiles (128 VGPRs, four workgroups per CU)
//   inverse, fused multiply-add: the loop is LDS-bound and 4096-sample tiles amortise the halo best
static const TileCfg kFwdCfgs[] = {{2, 1, 256}, {4, 1, 256}, {1, 1, 256}, {4, 2, 256}, {8, 2, 256}, {8, 4, 256}, {8, 10, 256}};
static const TileCfg kInvCfgs[] = {{4, 1, 256}, {2, 1, 256}, {8, 1, 256}, {1, 1, 256}, {4, 2, 256}, {8, 2, 256}, {8, 4, 256}, {8, 10, 256}};
static const TileCfg kInvCfgsFma[] = {{4, 1, 256}, {8, 1, 256}, {2, 1, 256}, {1, 1, 256}, {4, 2, 256}, {8, 2, 256}, {8, 4, 256}, {8, 10, 256}};

struct CfgList {
    const TileCfg* v;
    int n;
    bool inverse;
};
static CfgList fwd_cfgs() { return CfgList{kFwdCfgs, static_cast<int>(sizeof(kFwdCfgs) / sizeof(TileCfg)), false}; }
static CfgList inv_cfgs(bool fma) {
    const int n = static_cast<int>(sizeof(kInvCfgs) / sizeof(TileCfg));
    return CfgList{fma ? kInvCfgsFma : kInvCfgs, n, true};
}

// Plan option fwd_tile / inv_tile: force an instantiated tile shape (ignored when the halo does not fit it).
static bool forced_cfg(const PlanOptions& opt, const CfgList& cl, int hpt_need, TileCfg* out) {
    const int p = cl.inverse ? opt.inv_tile_p : opt.fwd_tile_p;
    const int h = cl.inverse ? opt.inv_tile_h : opt.fwd_tile_h;
    if (p <= 0 || h <= 0) return false;
    for (int i = 0; i < cl.n; ++i)
        if (cl.v[i].ppt == p && cl.v[i].hpt == h && h >= hpt_need) {
            *out = cl.v[i];
            return true;
        }
    return false;
}

static bool pick_cfg(const vw_ctx* ctx, int64_t n, int64_t halo, const CfgList& cl, TileCfg* out) {
    const int lds_limit = ctx->lds_limit;
    // two samples of slack: pair alignment may start a pair one sample outside the exact need
    int hpt_need = static_cast<int>((halo + 2 + kChunk - 1) / kChunk);
    if (hpt_need < 1) hpt_need = 1;
    if (forced_cfg(ctx->opt, cl, hpt_need, out)) return true;
    // smallest instantiated halo that covers the need and fits LDS
    int hpt = 0;
    for (int i = 0; i < cl.n; ++i) {
        const TileCfg& c = cl.v[i];
        if (c.hpt < hpt_need || c.lds_bytes(cl.inverse) > static_cast<size_t>(lds_limit)) continue;
        if (hpt == 0 || c.hpt < hpt) hpt = c.hpt;
    }
    if (hpt == 0) return false;
    // with that halo: the first shape in preference order whose tile does not exceed the row
    // (rounded up to a chunk), else the smallest tile
    const int64_t row_len = (n + kChunk - 1) / kChunk * kChunk;
    int best = -1, smallest = -1;
    for (int i = 0; i < cl.n; ++i) {
        const TileCfg& c = cl.v[i];
        if (c.hpt != hpt || c.lds_bytes(cl.inverse) > static_cast<size_t>(lds_limit)) continue;
        if (smallest < 0 || c.tile() < cl.v[smallest].tile()) smallest = i;
        if (best < 0 && c.tile() <= row_len) best = i;
    }
    *out = cl.v[best >= 0 ? best : smallest];
    return true;
}

// Left halo a fused forward group lev0..lev1 needs, and per-level start positions.
static int64_t fwd_halo(int lev0, int lev1, int lh, int lg, int64_t* hn /*[kMaxLevels+1]*/) {
    int64_t need = 0;  // Hn_{lev1} = 0
    if (hn != nullptr) hn[lev1] = 0;
    for (int j = lev1; j >= lev0; --j) {
        const int64_t s = static_cast<int64_t>(1) << (j - 1);
        const int64_t a = need + static_cast<int64_t>(lh - 1) * s;
        const int64_t b = static_cast<int64_t>(lg - 1) * s;
        need = a > b ? a : b;
        if (hn != nullptr && j - 1 >= 0) hn[j - 1] = need;
    }
    return need;
}

// Right halo a fused inverse group lev1..lev0 needs.
static int64_t inv_halo(int lev0, int lev1, int lh) {
    int64_t e = 0;
    for (int j = lev0; j <= lev1; ++j) e += static_cast<int64_t>(lh - 1) * (static_cast<int64_t>(1) << (j - 1));
    return e;
}

struct Step {
    int lev0, lev1;
    bool fused;
    TileCfg cfg;
    int sdry = 0;       // 0: tile kernels; 1, 2: dry-run-warmed streaming shapes (see plan_sdry)
    int ndry = 0;       // warm-up tiles per run
    int hist_cap = 0;   // doubles of history storage
    int32_t hoff[kMaxLevels + 2] = {0}, hlen[kMaxLevels + 2] = {0};
};

// at most one step per level; fixed storage so that planning never touches the heap
struct StepList {
    Step v[kMaxLevels + 1];
    int count = 0;
    void push_back(const Step& s) { v[count++] = s; }
    bool empty() const { return count == 0; }
    int size() const { return count; }
    const Step& operator[](int i) const { return v[i]; }
};

static void scale_taps(const double* src, int len, double* dst) {
    // MODWTTransform.java:133-142: scale = 1.0 / Math.sqrt(2.0); tap * scale
    const double scale = 1.0 / sqrt(2.0);
    for (int i = 0; i < len; ++i) dst[i] = src[i] * scale;
    for (int i = len; i < kMaxTaps + 2; ++i) dst[i] = 0.0;
}

// ---------------------------------------------------------------------------
// kernel-instance dispatch
// ---------------------------------------------------------------------------

// Streaming forward: split each row into runs of consecutive tiles, one workgroup per run.  Long runs
// amortise the one halo a run pays for; short runs keep enough workgroups in flight when the batch
// is small.  Plan option fwd_chunks caps the run length (1 = one tile per workgroup, as the halo kernels do).
static void plan_stream(const PlanOptions& opt, FwdArgs* a, int64_t rows, int lmax) {
    int off = 0;
    for (int j = a->lev0; j <= a->lev1; ++j) {
        const int64_t hl = (static_cast<int64_t>(lmax - 1) * (static_cast<int64_t>(1) << (j - 1)) + 1) & ~static_cast<int64_t>(1);
        if (hl > kChunk - 2 || off + hl > kChunk) return;  // history does not fit: keep one tile per workgroup
        a->hoff[j] = off;
        a->hlen[j] = static_cast<int32_t>(hl);
        off += static_cast<int>(hl);
    }
    // Runs of at most 4 tiles, all of one length where the tile count allows.  How fast this launch streams depends
    // on where the driver placed the buffers in physical memory (the same launch on the same rows takes 10.8 to
    // 13.5 ms memory-bound, in four discrete steps, depending only on what was allocated before the coefficient
    // block); long runs are the fastest on the best placement and the slowest on the worst (whole rows: 12.0 to
    // 13.9 ms in the reference's arithmetic), runs of 4 stay within 12.4 to 12.7 ms on all of them.  A ragged last
    // run costs more than a shorter run length (16 tiles as 5+5+5+1: 13.6 ms), hence the even deal.
    const int64_t cap = opt.fwd_chunks >= 1 ? opt.fwd_chunks : 4;
    const int64_t tiles = a->tiles_per_row;
    const int64_t want_wgs = opt.stream_wgs >= 1 ? opt.stream_wgs : 16384;  // several rounds of resident workgroups on 256 CUs
    int64_t nseg = rows > 0 ? (want_wgs + rows - 1) / rows : 1;
    if (nseg < (tiles + cap - 1) / cap) nseg = (tiles + cap - 1) / cap;
    if (nseg > tiles) nseg = tiles;
    if (nseg < 1) nseg = 1;
    a->ntiles = static_cast<int32_t>(nseg);
    a->run_base = static_cast<int32_t>(tiles / nseg);
    a->run_rem = static_cast<int32_t>(tiles % nseg);
    a->nchunks = a->run_base + (a->run_rem > 0 ? 1 : 0);
}

// Long-context streaming (k_fwd_sdry / k_inv_sdry): shape 1 = 512 lanes, 2048-sample tiles, histories up to 2046
// samples per level; shape 2 = 512 lanes, 4096-sample tiles, histories up to 2558.  Returns 0 if the
// level group does not fit either.
static int plan_sdry(const vw_ctx* ctx, int lev0, int lev1, int lmax, bool inverse, int32_t* hoff, int32_t* hlen, int* ndry,
                     int* hist_cap) {
    const int lds_limit = ctx->lds_limit;
    int64_t sum = 0, hmax = 0;
    for (int j = lev0; j <= lev1; ++j) {
        const int64_t hl = (static_cast<int64_t>(lmax - 1) * (static_cast<int64_t>(1) << (j - 1)) + 1) & ~static_cast<int64_t>(1);
        if (hl > 4096) return 0;
        hoff[j] = static_cast<int32_t>(sum);
        hlen[j] = static_cast<int32_t>(hl);
        sum += hl;
        if (hl > hmax) hmax = hl;
    }
    if (ctx->opt.no_sdry) return 0;
    // the inverse prefers the sixteen-wave shape when it fits: its LDS need (two buffers, two history sets) leaves room
    // for one workgroup per CU either way
    static const int kOrderFwd[] = {1, 2}, kOrderInv[] = {3, 1, 2};
    const int* order = inverse ? kOrderInv : kOrderFwd;
    const int norder = inverse ? 3 : 2;
    for (int oi = 0; oi < norder; ++oi) {
        const int v = order[oi];
        if (v == 3 && ctx->opt.no_wide_inv) continue;
        const SdryShape& sh = kSdry[v];
        const int64_t halo = static_cast<int64_t>(sh.hb) * kChunk;
        if (hmax > halo - 2 || hmax > sh.tile()) continue;
        const int64_t cap = (sum + 1023) / 1024 * 1024;
        // forward: one buffer and one history; inverse: R and W buffers, R and W histories
        const size_t lds = static_cast<size_t>(halo + sh.tile() + cap) * 8 * (inverse ? 2 : 1);
        if (lds > static_cast<size_t>(lds_limit)) continue;
        *ndry = static_cast<int>((sum + sh.tile() - 1) / sh.tile());
        *hist_cap = static_cast<int>(cap);
        return v;
    }
    return 0;
}

// Run length of the long-context kernels: long enough to amortise the warm-up tiles, short enough to
// fill the chip.  Plan option sdry_chunks overrides.
static void sdry_runs(const PlanOptions& opt, int64_t rows, int64_t tiles, int ndry, int32_t* nchunks, int32_t* nseg_out,
                      int64_t want_runs = 1024) {
    // about 1024 runs in all: two 512-lane workgroups fit a CU, and longer runs amortise the warm-up tiles better
    // than a third and fourth wave of workgroups would (sym8 J=8 on 64 x 777777: 24-48 tiles per run measured best)
    int64_t nseg = rows > 0 ? (want_runs + rows - 1) / rows : 1;
    if (nseg > tiles) nseg = tiles;
    if (nseg < 1) nseg = 1;
    int64_t nch = (tiles + nseg - 1) / nseg;
    const int64_t min_run = static_cast<int64_t>(ndry) * 6;
    if (nch < min_run) nch = min_run < tiles ? min_run : tiles;
    if (opt.sdry_chunks >= 1) nch = opt.sdry_chunks < tiles ? opt.sdry_chunks : tiles;
    if (nch < 1) nch = 1;
    *nchunks = static_cast<int32_t>(nch);
    *nseg_out = static_cast<int32_t>((tiles + nch - 1) / nch);
}

// Launches that move more than 1 GiB (four times the last-level cache) store their results with the streaming
// hint; smaller ones leave them cacheable for the kernel that reads them next.  Plan option nt_stores overrides.
static int streaming_stores(const PlanOptions& opt, int64_t b, int64_t n, int levels) {
    if (opt.nt_stores >= 0) return opt.nt_stores;
    return static_cast<double>(b) * static_cast<double>(n) * 8.0 * (levels + 2) > 1073741824.0;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------
// common argument checks
// ---------------------------------------------------------------------------

static int check_common(vw_ctx* ctx, int64_t b, int64_t n, int levels, int l0, int l1, int boundary) {
    if (ctx == nullptr) return VW_E_NULL;
    if (b < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative batch size %lld", static_cast<long long>(b));
    if (n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative signal length %lld", static_cast<long long>(n));
    if (boundary != VW_BOUNDARY_PERIODIC && boundary != VW_BOUNDARY_ZERO_PADDING &&
        boundary != VW_BOUNDARY_SYMMETRIC)
        return set_err(ctx, VW_E_BOUNDARY, "MODWT only supports PERIODIC, ZERO_PADDING, and SYMMETRIC boundary modes");
    if (l0 < 1 || l1 < 1 || l0 > kMaxTaps || l1 > kMaxTaps)
        return set_err(ctx, VW_E_LENGTH_MISMATCH, "filter lengths %d/%d outside [1, %d]", l0, l1, kMaxTaps);
    if (n == 0 && b > 0) return set_err(ctx, VW_E_EMPTY, "Signal cannot be empty");
    if (levels - 1 >= 31) return set_err(ctx, VW_E_TOO_LARGE, "Decomposition level would cause integer overflow");
    if (n > 0 && b > 0 && (b > INT64_MAX / n / (levels + 2))) return set_err(ctx, VW_E_TOO_LARGE, "batch too large");
    return VW_OK;
}

static int check_levels(vw_ctx* ctx, int64_t n, int levels, int lh, uint32_t flags) {
    if (flags & VW_FLAG_SINGLE_LEVEL) {
        if (levels != 1) return set_err(ctx, VW_E_LEVEL, "VW_FLAG_SINGLE_LEVEL requires levels == 1");
        return VW_OK;
    }
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of decomposition levels %d", levels);
    if (flags & VW_FLAG_NO_LEVEL_CHECK) return VW_OK;
    const int maxl = vw_max_levels(n, lh);
    if (levels > maxl)
        return set_err(ctx, VW_E_LEVEL, "Invalid number of decomposition levels %d (maximum %d for N=%lld, L=%d)", levels,
                       maxl, static_cast<long long>(n), lh);
    return VW_OK;
}

// number of taps of an L-tap filter whose dilated position l*s stays below n
// (MultiLevelMODWTTransform's truncation of a zero-stuffed filter to n entries)
static int truncated_taps(int len, int64_t s, int64_t n) {
    const int64_t keep = (n - 1) / s + 1;
    return keep < len ? static_cast<int>(keep) : len;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------

static int forward_dev(vw_ctx* ctx, const double* x, int64_t b, int64_t n, int levels, const double* h, int lh,
                       const double* g, int lg, int boundary, uint32_t flags, double* coeffs,
                       unsigned long long* first_bad_cell, hipStream_t stream) {
    Taps taps;
    scale_taps(h, lh, taps.lo);
    scale_taps(g, lg, taps.hi);
    const bool single = (flags & VW_FLAG_SINGLE_LEVEL) != 0;
    const int64_t band = n;
    const int64_t rowc = static_cast<int64_t>(levels + 1) * n;
    double* aslot = coeffs + static_cast<int64_t>(levels) * n;  // row 0 of the a_J band

    // truncation (multi-level API only) and boundary decide fused vs generic.  SYMMETRIC rides the tile-streaming
    // kernel only (one halo chunk, mirrored level by level in the first tile of a row): rows of at least one chunk,
    // level groups whose context fits one chunk
    const bool sym = (boundary == VW_BOUNDARY_SYMMETRIC);
    bool generic_all = sym && (n < kChunk || ctx->opt.no_sym_tiles);
    if (!single) {
        for (int j = 1; j <= levels; ++j) {
            const int64_t s = static_cast<int64_t>(1) << (j - 1);
            if (truncated_taps(lh, s, n) != lh || truncated_taps(lg, s, n) != lg) generic_all = true;
        }
    }
    auto pick = [&](int64_t hneed, TileCfg* out) {
        if (sym && hneed + 2 > kChunk) return false;
        if (!pick_cfg(ctx, n, hneed, fwd_cfgs(), out)) return false;
        return !sym || out->hpt == 1;
    };

    // ---- plan ----------------------------------------------------------------
    StepList steps;
    // a context beyond one chunk: all levels in one pass of the long-context streaming kernel if it fits
    if (!generic_all && !sym && fwd_halo(1, levels, lh, lg, nullptr) + 2 > kChunk) {
        Step st;
        st.lev0 = 1;
        st.lev1 = levels;
        st.fused = true;
        st.cfg = TileCfg{0, 0, kThreads};
        st.sdry = plan_sdry(ctx, 1, levels, lh > lg ? lh : lg, false, st.hoff, st.hlen, &st.ndry, &st.hist_cap);
        if (st.sdry != 0) steps.push_back(st);
    }
    if (steps.empty()) {
        int j = 1;
        while (j <= levels) {
            Step st;
            st.lev0 = j;
            st.lev1 = j;
            st.fused = false;
            st.cfg = TileCfg{0, 0, kThreads};
            if (!generic_all) {
                int hi = j;
                TileCfg cfg;
                if (pick(fwd_halo(j, j, lh, lg, nullptr), &cfg)) {
                    st.fused = true;
                    st.cfg = cfg;
                    while (hi + 1 <= levels) {
                        TileCfg c2;
                        const int64_t hneed = fwd_halo(j, hi + 1, lh, lg, nullptr);
                        if (hneed + 2 > static_cast<int64_t>(kHaloMaxChunks) * kChunk) break;
                        if (!pick(hneed, &c2)) break;
                        st.cfg = c2;
                        ++hi;
                    }
                    st.lev1 = hi;
                }
            }
            steps.push_back(st);
            j = st.lev1 + 1;
        }
    }
    const int nsteps = static_cast<int>(steps.size());
    double* wsp = nullptr;
    if (nsteps > 1) {
        const int rc = ensure_ws(ctx, static_cast<size_t>(b) * n * 8, &wsp);
        if (rc != VW_OK) return rc;
    }

    // ---- run -----------------------------------------------------------------
    const double* in = x;
    int64_t in_stride = n;
    bool first = true;
    ctx->plan_fwd[0] = 0;
    for (int i = 0; i < nsteps; ++i) {
        const Step& st = steps[i];
        plan_note(ctx->plan_fwd, st.sdry != 0 ? "k_fwd_sdry" : st.fused ? (st.cfg.hpt == 1 ? "k_fwd_stream" : "k_fwd_fused") : "k_level_fwd",
                  st.lev0, st.lev1);
        // ping-pong so that the last step lands in the a_J band
        const bool to_slot = ((nsteps - 1 - i) % 2) == 0;
        double* vout = to_slot ? aslot : wsp;
        const int64_t vstride = to_slot ? rowc : n;
        if (st.sdry != 0) {
            FwdArgs a;
            memset(&a, 0, sizeof(a));
            a.in = in;
            a.in_stride = in_stride;
            a.wout = coeffs;
            a.w_row_stride = rowc;
            a.w_band_stride = band;
            a.vout = vout;
            a.v_stride = vstride;
            a.n = n;
            a.n_store = n;
            a.nrows = b;
            a.lev0 = st.lev0;
            a.lev1 = st.lev1;
            a.lh = lh;
            a.lg = lg;
            a.ndry = st.ndry;
            a.hist_cap = st.hist_cap;
            memcpy(a.hoff, st.hoff, sizeof(a.hoff));
            memcpy(a.hlen, st.hlen, sizeof(a.hlen));
            const int T = kSdry[st.sdry].tile();
            const int64_t tiles = (n + T - 1) / T;
            a.tiles_per_row = static_cast<int32_t>(tiles);
            sdry_runs(ctx->opt, b, tiles, st.ndry, &a.nchunks, &a.ntiles);
            a.first_bad = first ? first_bad_cell : nullptr;
            a.taps = taps;
            a.nt_store = streaming_stores(ctx->opt, b, n, levels);
#ifdef VW_TUNING
            if (strstr(ctx->opt.ablate, "nostore")) a.n_store = 0;
            if (strstr(ctx->opt.ablate, "nocompute")) a.lh = a.lg = 0;
#endif
            const bool vec2 = (n % 2 == 0) && (in_stride % 2 == 0) && (vstride % 2 == 0) && aligned16(in) &&
                              aligned16(coeffs) && aligned16(vout);
            const int64_t nblocks = b * a.ntiles;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            hipError_t e = launch_fwd_sdry(a, st.sdry, boundary, vec2, (flags & VW_FLAG_FMA) != 0, ctx->opt.no_unroll == 0, nblocks, stream);
            if (e != hipSuccess) return set_err(ctx, VW_E_HIP, "forward kernel launch: %s", hipGetErrorString(e));
        } else if (st.fused) {
            FwdArgs a;
            memset(&a, 0, sizeof(a));
            a.in = in;
            a.in_stride = in_stride;
            a.wout = coeffs;
            a.w_row_stride = rowc;
            a.w_band_stride = band;
            a.vout = vout;
            a.v_stride = vstride;
            a.n = n;
            a.n_store = n;
            a.nrows = b;
            const int T = st.cfg.tile();
            a.ntiles = static_cast<int32_t>((n + T - 1) / T);
            a.tiles_per_row = a.ntiles;
            a.nchunks = 1;
            a.run_base = 1;
            a.run_rem = 0;
            a.lev0 = st.lev0;
            a.lev1 = st.lev1;
            a.lh = lh;
            a.lg = lg;
            if (st.cfg.hpt == 1) plan_stream(ctx->opt, &a, b, lh > lg ? lh : lg);
#ifdef VW_TUNING
            if (strstr(ctx->opt.ablate, "nostore")) a.n_store = 0;
            if (strstr(ctx->opt.ablate, "nocompute")) a.lh = a.lg = 0;
#endif
            int64_t hn[kMaxLevels + 2] = {0};
            fwd_halo(st.lev0, st.lev1, lh, lg, hn);
            const int64_t halo = static_cast<int64_t>(st.cfg.hpt) * kChunk;
            for (int j = st.lev0; j <= st.lev1; ++j) {
                int64_t ps = halo - hn[j];
                if (ps < 0) ps = 0;
                a.pstart[j] = static_cast<int32_t>(ps & ~static_cast<int64_t>(1));
#ifdef VW_TUNING
                if (strstr(ctx->opt.ablate, "nohalo")) a.pstart[j] = static_cast<int32_t>(halo);
#endif
            }
            a.first_bad = first ? first_bad_cell : nullptr;
            a.taps = taps;
            a.nt_store = streaming_stores(ctx->opt, b, n, levels);
            const bool vec2 = (n % 2 == 0) && (in_stride % 2 == 0) && (vstride % 2 == 0) && aligned16(in) &&
                              aligned16(coeffs) && aligned16(vout);
            const int64_t nblocks = b * a.ntiles;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            // (VW_FLAG_FMA permits fused rounding, it does not demand it: SYMMETRIC keeps the reference's arithmetic)
            hipError_t e = launch_fwd(a, st.cfg, boundary, vec2, (flags & VW_FLAG_FMA) != 0 && !sym, ctx->opt.no_unroll == 0, nblocks, stream);
            if (e != hipSuccess) return set_err(ctx, VW_E_HIP, "forward kernel launch: %s", hipGetErrorString(e));
        } else {
            if (first && first_bad_cell != nullptr) {
                hipLaunchKernelGGL(k_first_non_finite, dim3(1024), dim3(kThreads), 0, stream, x, b * n, first_bad_cell);
            }
            LevelArgs a;
            memset(&a, 0, sizeof(a));
            const int j = st.lev0;
            a.in = in;
            a.in_stride = in_stride;
            a.out_lo = vout;
            a.out_lo_stride = vstride;
            a.out_hi = coeffs + static_cast<int64_t>(j - 1) * band;
            a.out_hi_stride = rowc;
            a.n = n;
            a.nrows = b;
            a.stride = static_cast<int64_t>(1) << (j - 1);
            a.llo = single ? lh : truncated_taps(lh, a.stride, n);
            a.lhi = single ? lg : truncated_taps(lg, a.stride, n);
            a.taps = taps;
            const int64_t per_row = (n + kThreads - 1) / kThreads;
            const int64_t nblocks = per_row * b;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            if (boundary == VW_BOUNDARY_PERIODIC)
                hipLaunchKernelGGL(k_level_fwd<kPeriodic>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, stream, a);
            else if (boundary == VW_BOUNDARY_ZERO_PADDING)
                hipLaunchKernelGGL(k_level_fwd<kZeroPad>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, stream, a);
            else
                hipLaunchKernelGGL(k_level_fwd<kSymmetric>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, stream, a);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return set_err(ctx, VW_E_HIP, "forward level kernel launch: %s", hipGetErrorString(e));
        }
        in = vout;
        in_stride = vstride;
        first = false;
    }
    return VW_OK;
}

static int forward_checks(vw_ctx* ctx, const void* x, int64_t b, int64_t n, int levels, const double* h, int lh,
                          const double* g, int lg, int boundary, uint32_t flags, const void* coeffs) {
    if (ctx == nullptr) return VW_E_NULL;
    if (h == nullptr || g == nullptr) return set_err(ctx, VW_E_NULL, "filter cannot be null");
    int rc = check_common(ctx, b, n, levels, lh, lg, boundary);
    if (rc != VW_OK) return rc;
    if (b > 0 && (x == nullptr || coeffs == nullptr)) return set_err(ctx, VW_E_NULL, "signal cannot be null");
    return check_levels(ctx, n, levels, lh, flags);
}

extern "C" int32_t vw_modwt_forward_batch_dev(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int32_t levels,
                                              const double* h, int32_t lh, const double* g, int32_t lg,
                                              int32_t boundary, uint32_t flags, double* coeffs_dev,
                                              int64_t* first_bad_dev, void* stream) {
    int rc = forward_checks(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, flags, coeffs_dev);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long* cell = nullptr;
    if ((flags & VW_FLAG_CHECK_FINITE) && first_bad_dev != nullptr) {
        VW_HIP(ctx, hipMemsetAsync(first_bad_dev, 0xff, 8, s));
        cell = reinterpret_cast<unsigned long long*>(first_bad_dev);
    }
    return forward_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, flags, coeffs_dev, cell, s);
}

// Rows per staging pass of the host-pointer entry points.
static int64_t stage_rows(int64_t b, int64_t n, int bands_total) {
    const int64_t budget = static_cast<int64_t>(4) << 30;  // ~4 GiB of device staging per pass
    int64_t rows = budget / (n * 8 * bands_total);
    if (rows < 1) rows = 1;
    return rows < b ? rows : b;
}

extern "C" int32_t vw_modwt_forward_batch(vw_ctx* ctx, const double* x, int64_t b, int64_t n, int32_t levels,
                                          const double* h, int32_t lh, const double* g, int32_t lg, int32_t boundary,
                                          uint32_t flags, double* coeffs, int64_t* first_bad) {
    if (first_bad != nullptr) *first_bad = -1;
    int rc = forward_checks(ctx, x, b, n, levels, h, lh, g, lg, boundary, flags, coeffs);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t rows = stage_rows(b, n, levels + 2);
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(rows) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, static_cast<size_t>(rows) * (levels + 1) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->cell, 8);
        if (rc != VW_OK) return rc;
    }
    double* dx = static_cast<double*>(ctx->stage_in.p);
    double* dc = static_cast<double*>(ctx->stage_out.p);
    int64_t* cell = static_cast<int64_t*>(ctx->cell.p);
    const bool chk = (flags & VW_FLAG_CHECK_FINITE) != 0;
    for (int64_t r0 = 0; r0 < b; r0 += rows) {
        const int64_t nr = (b - r0) < rows ? (b - r0) : rows;
        VW_HIP(ctx, hipMemcpyAsync(dx, x + r0 * n, static_cast<size_t>(nr) * n * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = vw_modwt_forward_batch_dev(ctx, dx, nr, n, levels, h, lh, g, lg, boundary, flags | VW_FLAG_NO_LEVEL_CHECK,
                                        dc, chk ? cell : nullptr, ctx->stream);
        if (rc != VW_OK) return rc;
        VW_HIP(ctx, hipMemcpyAsync(coeffs + r0 * (levels + 1) * n, dc, static_cast<size_t>(nr) * (levels + 1) * n * 8,
                                   hipMemcpyDeviceToHost, ctx->stream));
        int64_t bad = -1;
        if (chk) VW_HIP(ctx, hipMemcpyAsync(&bad, cell, 8, hipMemcpyDeviceToHost, ctx->stream));
        VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (chk && bad >= 0) {
            if (first_bad != nullptr) *first_bad = r0 * n + bad;
            return set_err(ctx, VW_E_NONFINITE, "signal contains a non-finite value at index %lld",
                           static_cast<long long>(r0 * n + bad));
        }
    }
    return VW_OK;
}

// ---------------------------------------------------------------------------
// inverse
// ---------------------------------------------------------------------------

// SYMMETRIC: every level of the cascade goes through the left-looking tile kernel -- then a denoise applies its
// thresholds as that kernel stages the bands, instead of in a pass of its own over the coefficient block
static bool sym_inverse_all_tiled(const vw_ctx* ctx, int lhr, int lgr, int levels, int64_t n) {
    if (n < kChunk || ctx->opt.no_sym_tiles) return false;
    for (int j = 1; j <= levels; ++j)
        if (!inv_sym_supported(lhr, lgr, j, j, n)) return false;
    return true;
}

static int inverse_dev(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int levels, const double* hr, int lhr,
                       const double* gr, int lgr, int boundary, uint32_t flags, uint32_t band_mask, int thr_mode,
                       const double* thr_rows, double* xout, unsigned long long* first_bad_cell, hipStream_t stream,
                       int64_t thr_row_stride = 1, int64_t thr_level_stride = 0) {
    Taps taps;
    scale_taps(hr, lhr, taps.lo);
    scale_taps(gr, lgr, taps.hi);
    const bool single = (flags & VW_FLAG_SINGLE_LEVEL) != 0;
    const int64_t band = n;
    const int64_t rowc = static_cast<int64_t>(levels + 1) * n;
    const double* aslot = coeffs + static_cast<int64_t>(levels) * n;

    // SYMMETRIC: the left-looking tile kernel (k_inv_sym) for the multi-level form and for MODWTTransform.inverse;
    // inverseBatchOptimized's flavour (reflects t + l at the row end, pair sum) is the ordinary single-level tile
    // kernel with a mirrored right halo
    const bool sym = (boundary == VW_BOUNDARY_SYMMETRIC);
    const bool sym_right = sym && single && (flags & VW_FLAG_SYMMETRIC_BATCH_VARIANT) != 0;
    bool generic_all = sym && (n < kChunk || ctx->opt.no_sym_tiles);
    if (!single) {
        for (int j = 1; j <= levels; ++j) {
            const int64_t s = static_cast<int64_t>(1) << (j - 1);
            if (truncated_taps(lhr, s, n) != lhr || truncated_taps(lgr, s, n) != lgr) generic_all = true;
        }
    }
    if (thr_mode != 0 && (generic_all || (sym && (sym_right || !sym_inverse_all_tiled(ctx, lhr, lgr, levels, n)))))
        return set_err(ctx, VW_E_HIP, "internal: thresholded inverse needs the fused path");

    StepList steps;
    {
        int j = levels;
        while (j >= 1) {
            Step st;
            st.lev0 = j;
            st.lev1 = j;
            st.fused = false;
            st.cfg = TileCfg{0, 0, kThreads};
            // a context beyond one chunk: as many of the remaining levels as fit one pass of the
            // long-context streaming kernel
            if (!generic_all && sym && !sym_right) {
                // as many of the remaining levels as keep their joint left context inside one halo chunk
                if (inv_sym_supported(lhr, lgr, j, j, n)) {
                    int lo = j;
                    while (lo - 1 >= 1 && inv_sym_supported(lhr, lgr, lo - 1, j, n)) --lo;
                    st.lev0 = lo;
                    st.fused = true;
                    st.cfg = TileCfg{n >= 4 * kChunk ? 4 : 1, 1, kThreads};
                }
                steps.push_back(st);
                j = st.lev0 - 1;
                continue;
            }
            // (k_inv_sdry keeps 32-bit byte offsets into a row: rows below 2^29 samples)
            if (!generic_all && inv_halo(1, j, lhr) + 2 > kChunk && n < (static_cast<int64_t>(1) << 29)) {
                for (int lo = 1; lo <= j && st.sdry == 0; ++lo) {
                    st.sdry = plan_sdry(ctx, lo, j, lhr, true, st.hoff, st.hlen, &st.ndry, &st.hist_cap);
                    if (st.sdry != 0) st.lev0 = lo;
                }
                if (st.sdry != 0) {
                    st.fused = true;
                    steps.push_back(st);
                    j = st.lev0 - 1;
                    continue;
                }
            }
            if (!generic_all) {
                TileCfg cfg;
                if (pick_cfg(ctx, n, inv_halo(j, j, lhr), inv_cfgs((flags & VW_FLAG_FMA) != 0), &cfg) && (!sym || cfg.hpt == 1)) {
                    st.fused = true;
                    st.cfg = cfg;
                    int lo = j;
                    while (lo - 1 >= 1) {
                        TileCfg c2;
                        const int64_t hneed = inv_halo(lo - 1, j, lhr);
                        if (hneed + 2 > static_cast<int64_t>(kHaloMaxChunks) * kChunk) break;
                        if (!pick_cfg(ctx, n, hneed, inv_cfgs((flags & VW_FLAG_FMA) != 0), &c2)) break;
                        st.cfg = c2;
                        --lo;
                    }
                    st.lev0 = lo;
                }
            }
            steps.push_back(st);
            j = st.lev0 - 1;
        }
    }
    const int nsteps = static_cast<int>(steps.size());
    double* wsp = nullptr;
    if (nsteps > 1) {
        const int rc = ensure_ws(ctx, static_cast<size_t>(b) * n * 8, &wsp);
        if (rc != VW_OK) return rc;
    }

    const double* rin = aslot;
    int64_t rin_stride = rowc;
    bool use_rin = ((band_mask >> levels) & 1u) != 0;
    bool first = true;
    ctx->plan_inv[0] = 0;
    for (int i = 0; i < nsteps; ++i) {
        const Step& st = steps[i];
        const bool to_out = ((nsteps - 1 - i) % 2) == 0;
        double* rout = to_out ? xout : wsp;
        const int64_t rstride = n;
        if (st.sdry != 0) {
            InvArgs a;
            memset(&a, 0, sizeof(a));
            a.rin = rin;
            a.rin_stride = rin_stride;
            a.win = coeffs;
            a.w_row_stride = rowc;
            a.w_band_stride = band;
            a.rout = rout;
            a.rout_stride = rstride;
            a.n = n;
            a.n_store = n;
            a.nrows = b;
            a.lev0 = st.lev0;
            a.lev1 = st.lev1;
            a.lh = lhr;
            a.ndry = st.ndry;
            a.hist_cap = st.hist_cap;
            memcpy(a.hoff, st.hoff, sizeof(a.hoff));
            memcpy(a.hlen, st.hlen, sizeof(a.hlen));
            const int T = kSdry[st.sdry].tile();
            const int64_t tiles = (n + T - 1) / T;
            a.tiles_per_row = static_cast<int32_t>(tiles);
            // the sixteen-wave shape runs one workgroup per CU: half as many runs fill the chip equally often
            sdry_runs(ctx->opt, b, tiles, st.ndry, &a.nchunks, &a.ntiles, st.sdry == 3 ? 512 : 1024);
            a.band_mask = (band_mask & 0x7fffffffu) | (use_rin ? 0x80000000u : 0u);
            a.thr_mode = thr_mode;
            a.thr = thr_rows;
            a.thr_row_stride = thr_row_stride;
            a.thr_level_stride = thr_level_stride;
            a.first_bad = first_bad_cell;
            a.check_rin = first ? 1 : 0;
            a.taps = taps;
            a.nt_store = streaming_stores(ctx->opt, b, n, levels);
#ifdef VW_TUNING
            if (strstr(ctx->opt.ablate, "nostore")) a.n_store = 0;
            if (strstr(ctx->opt.ablate, "nocompute")) a.lh = 0;
            if (strstr(ctx->opt.ablate, "noload")) a.ablate |= 2;
#endif
            const bool vec2 = (n % 2 == 0) && (rin_stride % 2 == 0) && aligned16(rin) && aligned16(coeffs) && aligned16(rout);
            const int64_t nblocks = b * a.ntiles;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            plan_note(ctx->plan_inv, "k_inv_sdry", st.lev0, st.lev1);
            hipError_t e = launch_inv_sdry(a, st.sdry, boundary, ctx->opt.inv_vm >= 0 ? (ctx->opt.inv_vm == 1 && !vec2 ? 2 : ctx->opt.inv_vm) : (vec2 ? 1 : 2), (flags & VW_FLAG_FMA) != 0, ctx->opt.no_unroll == 0, nblocks, stream);
            if (e != hipSuccess) return set_err(ctx, VW_E_HIP, "inverse kernel launch: %s", hipGetErrorString(e));
        } else if (st.fused && sym && !sym_right) {
            InvArgs a;
            memset(&a, 0, sizeof(a));
            a.rin = rin;
            a.rin_stride = rin_stride;
            a.win = coeffs;
            a.w_row_stride = rowc;
            a.w_band_stride = band;
            a.rout = rout;
            a.rout_stride = rstride;
            a.n = n;
            a.n_store = n;
            a.nrows = b;
            const int T = st.cfg.tile();
            a.ntiles = static_cast<int32_t>((n + T - 1) / T);
            a.lev0 = st.lev0;
            a.lev1 = st.lev1;
            a.lh = lhr;
            a.lg = lgr;
            // first halo position from which R_j is valid: everything that was loaded for R_{lev1}, then the joint left
            // context of the levels undone so far
            const int lmax = lhr > lgr ? lhr : lgr;
            a.pend[st.lev1] = 0;
            for (int j = st.lev1; j > st.lev0; --j) {
                int64_t v = a.pend[j] + static_cast<int64_t>(lmax - 1) * (static_cast<int64_t>(1) << (j - 1));
                v = (v + 1) & ~static_cast<int64_t>(1);
                a.pend[j - 1] = static_cast<int32_t>(v < kChunk ? v : kChunk);
            }
            a.band_mask = (band_mask & 0x7fffffffu) | (use_rin ? 0x80000000u : 0u);
            a.thr_mode = thr_mode;
            a.thr = thr_rows;
            a.thr_row_stride = thr_row_stride;
            a.thr_level_stride = thr_level_stride;
            a.first_bad = first_bad_cell;
            a.check_rin = first ? 1 : 0;
            a.taps = taps;
            a.nt_store = streaming_stores(ctx->opt, b, n, levels);
            const bool vec2 = (n % 2 == 0) && (rin_stride % 2 == 0) && aligned16(rin) && aligned16(coeffs) && aligned16(rout) &&
                              !ctx->opt.no_vec2;
            const int64_t nblocks = b * a.ntiles;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            plan_note(ctx->plan_inv, "k_inv_sym", st.lev0, st.lev1);
            hipError_t er = launch_inv_sym(a, st.cfg.ppt, vec2, single, nblocks, stream);
            if (er != hipSuccess) return set_err(ctx, VW_E_HIP, "inverse kernel launch: %s", hipGetErrorString(er));
        } else if (st.fused) {
            InvArgs a;
            memset(&a, 0, sizeof(a));
            a.rin = rin;
            a.rin_stride = rin_stride;
            a.win = coeffs;
            a.w_row_stride = rowc;
            a.w_band_stride = band;
            a.rout = rout;
            a.rout_stride = rstride;
            a.n = n;
            a.n_store = n;
            a.nrows = b;
            const int T = st.cfg.tile();
            const int M = T + st.cfg.hpt * kChunk;
            a.ntiles = static_cast<int32_t>((n + T - 1) / T);
            a.lev0 = st.lev0;
            a.lev1 = st.lev1;
            a.lh = lhr;
            int64_t e = 0;  // E_{lev0-1} = 0
            a.pend[st.lev0 - 1] = T;
            for (int j = st.lev0; j <= st.lev1; ++j) {
                e += static_cast<int64_t>(lhr - 1) * (static_cast<int64_t>(1) << (j - 1));
                int64_t pe = T + ((e + 1) & ~static_cast<int64_t>(1));
                if (pe > M) pe = M;
                a.pend[j] = static_cast<int32_t>(pe);
            }
            a.band_mask = (band_mask & 0x7fffffffu) | (use_rin ? 0x80000000u : 0u);
            a.thr_mode = thr_mode;
            a.thr = thr_rows;
            a.thr_row_stride = thr_row_stride;
            a.thr_level_stride = thr_level_stride;
            a.first_bad = first_bad_cell;
            a.check_rin = first ? 1 : 0;
            a.taps = taps;
            a.nt_store = streaming_stores(ctx->opt, b, n, levels);
#ifdef VW_TUNING
            if (strstr(ctx->opt.ablate, "nostore")) a.n_store = 0;
            if (strstr(ctx->opt.ablate, "nocompute")) a.lh = 0;
            if (strstr(ctx->opt.ablate, "nohalo"))
                for (int j = st.lev0; j < st.lev1; ++j) a.pend[j] = T;
            if (strstr(ctx->opt.ablate, "ringnomath")) a.ablate |= 1;
            if (strstr(ctx->opt.ablate, "ringnoload")) a.ablate |= 2;
            if (strstr(ctx->opt.ablate, "ringonewg")) a.ablate |= 8;
#endif
            const bool vec2 = (n % 2 == 0) && (rin_stride % 2 == 0) && aligned16(rin) && aligned16(coeffs) && aligned16(rout) &&
                              !ctx->opt.no_vec2;
            const int64_t nblocks = b * a.ntiles;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            const bool blk = st.cfg.ppt == 4 && st.cfg.hpt == 1 && !ctx->opt.no_blk && inv_blk_supported(a.lh, st.lev1) && !sym;
            const bool ring = blk && vec2 && n >= 2560 && !ctx->opt.no_ring && ctx->lds_limit >= 81920;
            // opt-in: runs of tiles with carried right contexts
            bool carry = ring && ctx->opt.carry_chunks >= 2 && a.ntiles > 1 && inv_carry_supported(a.lh, st.lev0, st.lev1);
            int64_t nlaunch = nblocks;
            const bool plain_inv = thr_mode == 0 && first_bad_cell == nullptr && a.band_mask == 0xffffffffu;
            // one wavefront per run of chunks: wins once every CU holds its five waves several times over and the runs are
            // long enough for the warm-up chunk not to matter; below that the tile kernels stay
            // (a denoise that only thresholds goes along: the bands are thresholded as they land)
            const bool thr_only = thr_mode != 0 && !(flags & VW_FLAG_FMA) && first_bad_cell == nullptr && a.band_mask == 0xffffffffu;
            bool wave = ctx->opt.inv_wave >= 0 && vec2 && (plain_inv || thr_only) && inv_wave_supported(a.lh, st.lev0, st.lev1, boundary, n) &&
                        ctx->lds_limit >= 32768;
            if (wave) {
                const int64_t chunks = (n + 511) / 512;
                const int ndry = inv_wave_ndry(a.lh, st.lev1);
                int64_t nch;
                if (ctx->opt.inv_wave >= 1) {
                    nch = ctx->opt.inv_wave < chunks ? ctx->opt.inv_wave : chunks;
                    if (nch < ndry) nch = ndry;
                } else {
                    const int64_t want_waves = 16 * 5 * static_cast<int64_t>(ctx->num_cus);
                    int64_t runs = (want_waves + b - 1) / b;
                    if (runs > chunks / 16) runs = chunks / 16;
                    if (runs < 1) runs = 1;
                    nch = (chunks + runs - 1) / runs;
                    // shorter filters leave the tile kernels bound by memory alone, and the tile order streams faster
                    // (measured on 65536 x 16384, inverse: Haar 11.8 ms ring / 13.8 ms wave, DB2 12.2 / 13.6, DB4 15.7 / 13.8)
                    if (b * chunks < 4 * want_waves || chunks < 8 || a.lh < 8) wave = false;
                }
                if (wave) {
                    a.tiles_per_row = static_cast<int32_t>(chunks);
                    a.nchunks = static_cast<int32_t>(nch);
                    a.ntiles = static_cast<int32_t>((chunks + nch - 1) / nch);
                    a.ndry = ndry;
                    a.hoff[0] = ctx->opt.inv_wave_nt >= 0 ? ctx->opt.inv_wave_nt : a.nt_store;
                    nlaunch = b * a.ntiles;
                    if (nlaunch > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
                    carry = false;
                }
            }
            if (carry) {
                const int64_t tiles = a.ntiles;
                const int64_t nch = ctx->opt.carry_chunks < tiles ? ctx->opt.carry_chunks : tiles;
                a.tiles_per_row = static_cast<int32_t>(tiles);
                a.nchunks = static_cast<int32_t>(nch);
                a.ntiles = static_cast<int32_t>((tiles + nch - 1) / nch);
                nlaunch = b * a.ntiles;
            }
            plan_note(ctx->plan_inv, wave ? "k_inv_wave" : carry ? "k_inv_carry" : ring ? "k_inv_ring" : blk ? "k_inv_blk" : "k_inv_fused", st.lev0, st.lev1);
            hipError_t er = wave ? launch_inv_wave(a, boundary, (flags & VW_FLAG_FMA) != 0, nlaunch, stream)
                            : carry ? launch_inv_carry(a, boundary, (flags & VW_FLAG_FMA) != 0, nlaunch, stream)
                            : ring ? launch_inv_ring(a, boundary, (flags & VW_FLAG_FMA) != 0, nblocks, stream)
                            : blk ? launch_inv_blk(a, boundary, vec2, (flags & VW_FLAG_FMA) != 0, nblocks, stream)
                                  : launch_inv(a, st.cfg, boundary, vec2, (flags & VW_FLAG_FMA) != 0 && !sym, nblocks, stream);
            if (er != hipSuccess) return set_err(ctx, VW_E_HIP, "inverse kernel launch: %s", hipGetErrorString(er));
        } else {
            if (first && first_bad_cell != nullptr)
                hipLaunchKernelGGL(k_first_non_finite, dim3(1024), dim3(kThreads), 0, stream, coeffs, b * rowc, first_bad_cell);
            plan_note(ctx->plan_inv, "k_level_inv", st.lev0, st.lev1);
            LevelArgs a;
            memset(&a, 0, sizeof(a));
            const int j = st.lev0;
            a.in = rin;
            a.in_stride = rin_stride;
            a.din = coeffs + static_cast<int64_t>(j - 1) * band;
            a.din_stride = rowc;
            a.out_lo = rout;
            a.out_lo_stride = rstride;
            a.n = n;
            a.nrows = b;
            a.stride = static_cast<int64_t>(1) << (j - 1);
            a.llo = single ? lhr : truncated_taps(lhr, a.stride, n);
            a.lhi = single ? lgr : truncated_taps(lgr, a.stride, n);
            a.variant = single ? ((flags & VW_FLAG_SYMMETRIC_BATCH_VARIANT) ? 2 : 1) : 0;
            a.use_in = use_rin ? 1 : 0;
            a.use_din = ((band_mask >> (j - 1)) & 1u) ? 1 : 0;
            // a level whose context fits no tile kernel inside a thresholding plan (e.g. 22 taps at level 9): W_j is
            // thresholded as this kernel reads it, like the tile kernels on either side do
            a.thr_mode = thr_mode;
            a.thr = thr_rows;
            a.thr_row_stride = thr_row_stride;
            a.thr_off = static_cast<int64_t>(j - 1) * thr_level_stride;
            a.taps = taps;
            const int64_t per_row = (n + kThreads - 1) / kThreads;
            const int64_t nblocks = per_row * b;
            if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
            if (boundary == VW_BOUNDARY_PERIODIC)
                hipLaunchKernelGGL(k_level_inv<kPeriodic>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, stream, a);
            else if (boundary == VW_BOUNDARY_ZERO_PADDING)
                hipLaunchKernelGGL(k_level_inv<kZeroPad>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, stream, a);
            else
                hipLaunchKernelGGL(k_level_inv<kSymmetric>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, stream, a);
            hipError_t er = hipGetLastError();
            if (er != hipSuccess) return set_err(ctx, VW_E_HIP, "inverse level kernel launch: %s", hipGetErrorString(er));
        }
        rin = rout;
        rin_stride = rstride;
        use_rin = true;
        first = false;
    }
    return VW_OK;
}

static int inverse_checks(vw_ctx* ctx, const void* coeffs, int64_t b, int64_t n, int levels, const double* hr, int lhr,
                          const double* gr, int lgr, int boundary, uint32_t flags, const void* x) {
    if (ctx == nullptr) return VW_E_NULL;
    if (hr == nullptr || gr == nullptr) return set_err(ctx, VW_E_NULL, "filter cannot be null");
    int rc = check_common(ctx, b, n, levels, lhr, lgr, boundary);
    if (rc != VW_OK) return rc;
    if (b > 0 && (x == nullptr || coeffs == nullptr)) return set_err(ctx, VW_E_NULL, "coefficients cannot be null");
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of levels %d", levels);
    if ((flags & VW_FLAG_SINGLE_LEVEL) && levels != 1) return set_err(ctx, VW_E_LEVEL, "VW_FLAG_SINGLE_LEVEL requires levels == 1");
    // The reference loops l < len(lowPassRecon) and indexes highPassRecon[l].
    const bool pair_loop = boundary != VW_BOUNDARY_SYMMETRIC || (flags & VW_FLAG_SINGLE_LEVEL);
    if (pair_loop && lgr < lhr)
        return set_err(ctx, VW_E_LENGTH_MISMATCH,
                       "high-pass reconstruction filter (%d taps) shorter than low-pass (%d): the reference indexes out of bounds here",
                       lgr, lhr);
    return VW_OK;
}

extern "C" int32_t vw_modwt_inverse_bands_batch_dev(vw_ctx* ctx, const double* coeffs_dev, int64_t b, int64_t n,
                                                    int32_t levels, const double* hr, int32_t lhr, const double* gr,
                                                    int32_t lgr, int32_t boundary, uint32_t band_mask, double* x_dev,
                                                    void* stream) {
    int rc = inverse_checks(ctx, coeffs_dev, b, n, levels, hr, lhr, gr, lgr, boundary, 0, x_dev);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    return inverse_dev(ctx, coeffs_dev, b, n, levels, hr, lhr, gr, lgr, boundary, 0, band_mask, 0, nullptr, x_dev,
                       nullptr, s);
}

extern "C" int32_t vw_modwt_inverse_batch_dev(vw_ctx* ctx, const double* coeffs_dev, int64_t b, int64_t n,
                                              int32_t levels, const double* hr, int32_t lhr, const double* gr,
                                              int32_t lgr, int32_t boundary, uint32_t flags, double* x_dev,
                                              int64_t* first_bad_dev, void* stream) {
    int rc = inverse_checks(ctx, coeffs_dev, b, n, levels, hr, lhr, gr, lgr, boundary, flags, x_dev);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned long long* cell = nullptr;
    if ((flags & VW_FLAG_CHECK_FINITE) && first_bad_dev != nullptr) {
        VW_HIP(ctx, hipMemsetAsync(first_bad_dev, 0xff, 8, s));
        cell = reinterpret_cast<unsigned long long*>(first_bad_dev);
    }
    return inverse_dev(ctx, coeffs_dev, b, n, levels, hr, lhr, gr, lgr, boundary, flags, 0xffffffffu, 0, nullptr, x_dev,
                       cell, s);
}

static int inverse_host(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int32_t levels, const double* hr,
                        int32_t lhr, const double* gr, int32_t lgr, int32_t boundary, uint32_t flags,
                        uint32_t band_mask, bool use_mask, double* x, int64_t* first_bad) {
    if (first_bad != nullptr) *first_bad = -1;
    int rc = inverse_checks(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, flags, x);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t rows = stage_rows(b, n, levels + 2);
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(rows) * (levels + 1) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, static_cast<size_t>(rows) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->cell, 8);
        if (rc != VW_OK) return rc;
    }
    double* dc = static_cast<double*>(ctx->stage_in.p);
    double* dx = static_cast<double*>(ctx->stage_out.p);
    int64_t* cell = static_cast<int64_t*>(ctx->cell.p);
    const bool chk = (flags & VW_FLAG_CHECK_FINITE) != 0;
    for (int64_t r0 = 0; r0 < b; r0 += rows) {
        const int64_t nr = (b - r0) < rows ? (b - r0) : rows;
        VW_HIP(ctx, hipMemcpyAsync(dc, coeffs + r0 * (levels + 1) * n, static_cast<size_t>(nr) * (levels + 1) * n * 8,
                                   hipMemcpyHostToDevice, ctx->stream));
        if (use_mask)
            rc = vw_modwt_inverse_bands_batch_dev(ctx, dc, nr, n, levels, hr, lhr, gr, lgr, boundary, band_mask, dx, ctx->stream);
        else
            rc = vw_modwt_inverse_batch_dev(ctx, dc, nr, n, levels, hr, lhr, gr, lgr, boundary, flags, dx,
                                            chk ? cell : nullptr, ctx->stream);
        if (rc != VW_OK) return rc;
        VW_HIP(ctx, hipMemcpyAsync(x + r0 * n, dx, static_cast<size_t>(nr) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        int64_t bad = -1;
        if (chk && !use_mask) VW_HIP(ctx, hipMemcpyAsync(&bad, cell, 8, hipMemcpyDeviceToHost, ctx->stream));
        VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (chk && !use_mask && bad >= 0) {
            if (first_bad != nullptr) *first_bad = r0 * (levels + 1) * n + bad;
            return set_err(ctx, VW_E_NONFINITE, "MODWTResult contains invalid coefficients (index %lld)",
                           static_cast<long long>(r0 * (levels + 1) * n + bad));
        }
    }
    return VW_OK;
}

extern "C" int32_t vw_modwt_inverse_batch(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int32_t levels,
                                          const double* hr, int32_t lhr, const double* gr, int32_t lgr,
                                          int32_t boundary, uint32_t flags, double* x, int64_t* first_bad) {
    return inverse_host(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, flags, 0xffffffffu, false, x, first_bad);
}

extern "C" int32_t vw_modwt_inverse_bands_batch(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n,
                                                int32_t levels, const double* hr, int32_t lhr, const double* gr,
                                                int32_t lgr, int32_t boundary, uint32_t band_mask, double* x) {
    return inverse_host(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, 0, band_mask, true, x, nullptr);
}

// ---------------------------------------------------------------------------
// thresholds / universal threshold / denoise
// ---------------------------------------------------------------------------

extern "C" int32_t vw_threshold_dev(vw_ctx* ctx, const double* in_dev, int64_t count, double threshold, int32_t soft,
                                    double* out_dev, void* stream) {
    if (ctx == nullptr) return VW_E_NULL;
    if (count < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative count");
    if (count > 0 && (in_dev == nullptr || out_dev == nullptr)) return set_err(ctx, VW_E_NULL, "Coefficients array cannot be null");
    if (threshold < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "Threshold must be non-negative");  // NaN passes, as in the reference
    if (count == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t blocks = (count + kThreads - 1) / kThreads;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_threshold, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, in_dev, out_dev, count,
                       threshold, soft ? 3 : 2);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_threshold(vw_ctx* ctx, const double* in, int64_t count, double threshold, int32_t soft,
                                double* out) {
    if (ctx == nullptr) return VW_E_NULL;
    if (count < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative count");
    if (count > 0 && (in == nullptr || out == nullptr)) return set_err(ctx, VW_E_NULL, "Coefficients array cannot be null");
    if (threshold < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "Threshold must be non-negative");
    if (count == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        int rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(count) * 8);
        if (rc != VW_OK) return rc;
    }
    double* d = static_cast<double*>(ctx->stage_in.p);
    VW_HIP(ctx, hipMemcpyAsync(d, in, static_cast<size_t>(count) * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = vw_threshold_dev(ctx, d, count, threshold, soft, d, ctx->stream);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(out, d, static_cast<size_t>(count) * 8, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

static double host_sqrt_2_ln_n(int64_t n) { return sqrt(2 * log(static_cast<double>(n))); }

// keys per lane of the register-resident selection kernels for a row of n samples, 0 when it is too long
static int reg_kpl(const PlanOptions& opt, int64_t n) {
    if (opt.no_reg_select) return 0;
    if (n <= 256) return 4;
    if (n <= 512) return 8;
    if (n <= 1024) return 16;
    if (n <= 2048) return 32;
    return 0;
}

static hipError_t launch_universal_threshold(const vw_ctx* ctx, const MedianArgs& a, int64_t b, int64_t n, size_t lds, hipStream_t s) {
    const dim3 grid(static_cast<unsigned>(b));
    switch (reg_kpl(ctx->opt, n)) {
        case 4: hipLaunchKernelGGL(k_universal_threshold_reg<4>, grid, dim3(64), 0, s, a); return hipGetLastError();
        case 8: hipLaunchKernelGGL(k_universal_threshold_reg<8>, grid, dim3(64), 0, s, a); return hipGetLastError();
        case 16: hipLaunchKernelGGL(k_universal_threshold_reg<16>, grid, dim3(64), 0, s, a); return hipGetLastError();
        case 32: hipLaunchKernelGGL(k_universal_threshold_reg<32>, grid, dim3(64), 0, s, a); return hipGetLastError();
        default: break;
    }
    if (!ctx->opt.no_wg_select && n > 2048 && n <= 32768) {
        // the row in the registers of one workgroup, read from HBM once
        if (n <= 4096)
            hipLaunchKernelGGL((k_universal_threshold_wg<256, 16>), grid, dim3(256), 0, s, a);
        else if (n <= 8192)
            hipLaunchKernelGGL((k_universal_threshold_wg<1024, 8>), grid, dim3(1024), 0, s, a);
        else if (n <= 16384)
            hipLaunchKernelGGL((k_universal_threshold_wg<1024, 16>), grid, dim3(1024), 0, s, a);
        else
            hipLaunchKernelGGL((k_universal_threshold_wg<1024, 32>), grid, dim3(1024), 0, s, a);
        return hipGetLastError();
    }
    return launch_tile_nt(k_universal_threshold<kThreads>, a, kThreads, b, lds, s);  // also reports a refused LDS size
}

// Longest row the 256-lane selection kernel copies into LDS, so that the nine passes of the radix select never leave
// the CU: whatever one workgroup can have beside the kernel's static 1.1 KiB (18432 samples = 144 KiB on gfx950).  Rows
// of 16384 samples took 3.07 ms per 16384 rows reading global memory nine times.
static int select_lds_cap(const vw_ctx* ctx) {
    int cap = (ctx->lds_limit - 16384) / 8;
    if (cap < 4096) cap = 4096;
    if (ctx->opt.select_lds_cap > 0 && ctx->opt.select_lds_cap < cap) cap = ctx->opt.select_lds_cap;
    return cap;
}

static int universal_threshold_dev(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int levels,
                                   double sqrt_2_ln_n, double* thr, hipStream_t s) {
    MedianArgs a;
    a.d1 = coeffs;
    a.row_stride = static_cast<int64_t>(levels + 1) * n;
    a.n = n;
    a.nrows = b;
    a.sqrt_2_ln_n = sqrt_2_ln_n < 0 ? host_sqrt_2_ln_n(n) : sqrt_2_ln_n;
    a.thr_out = thr;
    a.lds_cap = select_lds_cap(ctx);
    const size_t lds = (n <= a.lds_cap) ? static_cast<size_t>((n + 1) & ~static_cast<int64_t>(1)) * 8 : 16;
    if (b > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    VW_HIP(ctx, launch_universal_threshold(ctx, a, b, n, lds, s));
    return VW_OK;
}

extern "C" int32_t vw_swt_universal_threshold_batch_dev(vw_ctx* ctx, const double* coeffs_dev, int64_t b, int64_t n,
                                                        int32_t levels, double sqrt_2_ln_n, double* thresholds_dev,
                                                        void* stream) {
    if (ctx == nullptr) return VW_E_NULL;
    if (b < 0 || n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (b > 0 && (coeffs_dev == nullptr || thresholds_dev == nullptr)) return set_err(ctx, VW_E_NULL, "Result cannot be null");
    if (b > 0 && n == 0) return set_err(ctx, VW_E_EMPTY, "Signal cannot be empty");
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of levels %d", levels);
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    return universal_threshold_dev(ctx, coeffs_dev, b, n, levels, sqrt_2_ln_n, thresholds_dev, s);
}

extern "C" int32_t vw_swt_universal_threshold_batch(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n,
                                                    int32_t levels, double sqrt_2_ln_n, double* thresholds) {
    if (ctx == nullptr) return VW_E_NULL;
    if (b < 0 || n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (b > 0 && (coeffs == nullptr || thresholds == nullptr)) return set_err(ctx, VW_E_NULL, "Result cannot be null");
    if (b > 0 && n == 0) return set_err(ctx, VW_E_EMPTY, "Signal cannot be empty");
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of levels %d", levels);
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    // only band d_1 of each row is needed on the device
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        int rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(b) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, static_cast<size_t>(b) * 8);
        if (rc != VW_OK) return rc;
    }
    double* d = static_cast<double*>(ctx->stage_in.p);
    double* t = static_cast<double*>(ctx->stage_out.p);
    VW_HIP(ctx, hipMemcpy2DAsync(d, static_cast<size_t>(n) * 8, coeffs, static_cast<size_t>(levels + 1) * n * 8,
                                 static_cast<size_t>(n) * 8, static_cast<size_t>(b), hipMemcpyHostToDevice, ctx->stream));
    MedianArgs a;
    a.d1 = d;
    a.row_stride = n;
    a.n = n;
    a.nrows = b;
    a.sqrt_2_ln_n = sqrt_2_ln_n < 0 ? host_sqrt_2_ln_n(n) : sqrt_2_ln_n;
    a.thr_out = t;
    a.lds_cap = select_lds_cap(ctx);
    const size_t lds = (n <= a.lds_cap) ? static_cast<size_t>((n + 1) & ~static_cast<int64_t>(1)) * 8 : 16;
    VW_HIP(ctx, launch_universal_threshold(ctx, a, b, n, lds, ctx->stream));
    VW_HIP(ctx, hipMemcpyAsync(thresholds, t, static_cast<size_t>(b) * 8, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

extern "C" size_t vw_swt_denoise_workspace_bytes(int64_t b, int64_t n, int32_t levels) {
    if (b <= 0 || n <= 0 || levels < 1) return 0;
    // coefficient block + per-row thresholds (rounded up to 256 B)
    const size_t coeffs = static_cast<size_t>(b) * static_cast<size_t>(levels + 1) * static_cast<size_t>(n) * 8;
    const size_t thr = (static_cast<size_t>(b) * 8 + 255) & ~static_cast<size_t>(255);
    return coeffs + thr;
}

__global__ void __launch_bounds__(kThreads) k_fill(double* p, int64_t count, double v) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < count) p[i] = v;
}

// thresholds every detail band of every row in place (generic-path denoise)
__global__ void __launch_bounds__(kThreads) k_threshold_rows(double* coeffs, int64_t n, int levels,
                                                             const double* __restrict__ thr, int64_t thr_row_stride,
                                                             int64_t thr_level_stride, int mode) {
    const int64_t row = blockIdx.y;
    const int64_t rowc = static_cast<int64_t>(levels + 1) * n;
    const int64_t cnt = static_cast<int64_t>(levels) * n;
    double* p = coeffs + row * rowc;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < cnt;
         i += static_cast<int64_t>(gridDim.x) * kThreads)
        p[i] = apply_thr(p[i], thr[row * thr_row_stride + (i / n) * thr_level_stride], mode);
}

static size_t round256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

// ---- row-resident SWT denoise (k_swt_denoise_row): one launch, no coefficient block ------------------------------

static size_t row_denoise_lds(int64_t n, int levels, int lt) {
    const int64_t np = (n + 1) & ~static_cast<int64_t>(1);
    const int64_t hm = ((((static_cast<int64_t>(lt) - 1) << (levels - 1)) + 1) & ~static_cast<int64_t>(1)) + 2;
    return static_cast<size_t>(2 * (hm + np + hm) + static_cast<int64_t>(levels) * (np + hm)) * 8;
}

static bool row_denoise_eligible(const vw_ctx* ctx, int64_t n, int levels, int lh, int lg, int lhr, int lgr, int boundary) {
    if (ctx->opt.no_row_denoise) return false;
    if (boundary != VW_BOUNDARY_PERIODIC && boundary != VW_BOUNDARY_ZERO_PADDING) return false;
    if (!(lh == 2 || lh == 4 || lh == 8) || lg != lh || lhr != lh || lgr < lhr) return false;
    if (n < 2 || n > 2048 || levels < 1 || levels > vw_max_levels(n, lh)) return false;
    const size_t lds = row_denoise_lds(n, levels, lh);
    return lds <= 65536 && lds <= static_cast<size_t>(ctx->lds_limit);
}

template <int LT, int MODE>
static void launch_row_denoise_k(const RowDenoiseArgs& a, int64_t n, int64_t b, size_t lds, bool bands_in_regs, hipStream_t s) {
    const dim3 grid(static_cast<unsigned>(b)), block(64);
    if (bands_in_regs)
        hipLaunchKernelGGL((k_swt_denoise_rowreg<LT, MODE>), grid, block, lds, s, a);
    else if (n <= 256)
        hipLaunchKernelGGL((k_swt_denoise_row<LT, 4, MODE>), grid, block, lds, s, a);
    else if (n <= 512)
        hipLaunchKernelGGL((k_swt_denoise_row<LT, 8, MODE>), grid, block, lds, s, a);
    else if (n <= 1024)
        hipLaunchKernelGGL((k_swt_denoise_row<LT, 16, MODE>), grid, block, lds, s, a);
    else
        hipLaunchKernelGGL((k_swt_denoise_row<LT, 32, MODE>), grid, block, lds, s, a);
}

// rule 0: one threshold per row (fixed when threshold >= 0, else the SWT adapter's universal threshold); rule 1:
// WaveletDenoiser's per-level thresholds from sigma (method UNIVERSAL or MINIMAX), thresholds_out_dev [b][levels]
struct RowRule {
    int rule = 0, method = 0, level_scaled = 0;
    double log_n = 0.0;
};
static int row_denoise_dev(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int levels, const double* h, int lh, const double* g,
                           const double* hr, const double* gr, int boundary, double threshold, double sqrt_2_ln_n, int thr_mode,
                           double* y_dev, double* thresholds_out_dev, hipStream_t s, const RowRule& rr = RowRule()) {
    if (b > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    RowDenoiseArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x_dev;
    a.x_stride = n;
    a.y = y_dev;
    a.y_stride = n;
    a.n = n;
    a.nrows = b;
    a.levels = levels;
    a.thr_mode = thr_mode;
    a.threshold = threshold;
    a.sqrt_2_ln_n = sqrt_2_ln_n < 0 ? host_sqrt_2_ln_n(n) : sqrt_2_ln_n;
    a.thr_out = thresholds_out_dev;
    a.rule = rr.rule;
    a.method = rr.method;
    a.level_scaled = rr.level_scaled;
    a.log_n = rr.log_n;
    a.vec = (n % 2 == 0) && aligned16(x_dev) && aligned16(y_dev);
    scale_taps(h, lh, a.dec.lo);
    scale_taps(g, lh, a.dec.hi);
    scale_taps(hr, lh, a.rec.lo);
    scale_taps(gr, lh, a.rec.hi);
    a.taps_small = 1;
    for (int l = 0; l < lh; ++l)
        for (const double t : {a.dec.lo[l], a.dec.hi[l], a.rec.lo[l], a.rec.hi[l]})
            if (!(fabs(t) <= 0x1p20)) a.taps_small = 0;
    // rows of up to 512 samples at up to four levels keep their detail bands in registers (one band buffer in LDS)
    const bool regs = n <= 512 && levels <= 4 && !ctx->opt.no_band_regs;
    ctx->plan_fwd[0] = ctx->plan_inv[0] = 0;
    plan_note(ctx->plan_fwd, regs ? "k_swt_denoise_rowreg" : "k_swt_denoise_row", 1, levels);
    plan_note(ctx->plan_inv, regs ? "k_swt_denoise_rowreg" : "k_swt_denoise_row", 1, levels);
    const int64_t np = (n + 1) & ~static_cast<int64_t>(1);
    const int64_t hm = ((((static_cast<int64_t>(lh) - 1) << (levels - 1)) + 1) & ~static_cast<int64_t>(1)) + 2;
    const size_t lds = regs ? static_cast<size_t>(hm + np + hm + np + hm) * 8 : row_denoise_lds(n, levels, lh);
    const bool per = boundary == VW_BOUNDARY_PERIODIC;
    switch (lh) {
        case 2: per ? launch_row_denoise_k<2, kPeriodic>(a, n, b, lds, regs, s) : launch_row_denoise_k<2, kZeroPad>(a, n, b, lds, regs, s); break;
        case 4: per ? launch_row_denoise_k<4, kPeriodic>(a, n, b, lds, regs, s) : launch_row_denoise_k<4, kZeroPad>(a, n, b, lds, regs, s); break;
        default: per ? launch_row_denoise_k<8, kPeriodic>(a, n, b, lds, regs, s) : launch_row_denoise_k<8, kZeroPad>(a, n, b, lds, regs, s); break;
    }
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

static int swt_denoise_run(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int32_t levels, const double* h, int32_t lh,
                           const double* g, int32_t lg, const double* hr, int32_t lhr, const double* gr, int32_t lgr,
                           int32_t boundary, double threshold, double sqrt_2_ln_n, int32_t soft, double* y_dev,
                           double* thresholds_out_dev, char* wsb, hipStream_t s);

extern "C" int32_t vw_swt_denoise_batch_dev(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int32_t levels,
                                            const double* h, int32_t lh, const double* g, int32_t lg, const double* hr,
                                            int32_t lhr, const double* gr, int32_t lgr, int32_t boundary,
                                            double threshold, double sqrt_2_ln_n, int32_t soft, double* y_dev,
                                            double* thresholds_out_dev, void* workspace_dev, size_t workspace_bytes,
                                            void* stream) {
    int rc = forward_checks(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, 0, y_dev);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x_dev, b, n, levels, hr, lhr, gr, lgr, boundary, 0, y_dev);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t need = vw_swt_denoise_workspace_bytes(b, n, levels);
    if (workspace_dev != nullptr && workspace_bytes < need)
        return set_err(ctx, VW_E_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    // short rows: the whole decomposition of a row stays in LDS, no workspace is touched
    if (row_denoise_eligible(ctx, n, levels, lh, lg, lhr, lgr, boundary))
        return row_denoise_dev(ctx, x_dev, b, n, levels, h, lh, g, hr, gr, boundary, threshold, sqrt_2_ln_n, soft ? 1 : 2, y_dev,
                               thresholds_out_dev, s);
    char* wsb;
    if (workspace_dev != nullptr) {
        wsb = static_cast<char*>(workspace_dev);
    } else {
        rc = ensure(ctx, ctx->denoise_ws, need);
        if (rc != VW_OK) return rc;
        wsb = static_cast<char*>(ctx->denoise_ws.p);
    }
    return swt_denoise_run(ctx, x_dev, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold, sqrt_2_ln_n, soft, y_dev,
                           thresholds_out_dev, wsb, s);
}

// Entry point that allocates nothing (no counterpart in the reference).  Same kernels and results as
// vw_swt_denoise_batch_dev; the coefficient block, the thresholds and -- for plans of more than one pass -- the
// intermediate plane live in the static per-GPU scratch array instead of context-owned or caller-owned memory.
extern "C" int32_t vw_swt_denoise_batch_noalloc_dev(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int32_t levels,
                                                    const double* h, int32_t lh, const double* g, int32_t lg, const double* hr,
                                                    int32_t lhr, const double* gr, int32_t lgr, int32_t boundary,
                                                    double threshold, double sqrt_2_ln_n, int32_t soft, double* y_dev,
                                                    double* thresholds_out_dev, void* stream) {
    int rc = forward_checks(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, 0, y_dev);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x_dev, b, n, levels, hr, lhr, gr, lgr, boundary, 0, y_dev);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    if (row_denoise_eligible(ctx, n, levels, lh, lg, lhr, lgr, boundary)) {
        // short rows need no scratch at all: the decomposition of a row never leaves LDS
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        VW_HIP(ctx, hipSetDevice(ctx->device));
        return row_denoise_dev(ctx, x_dev, b, n, levels, h, lh, g, hr, gr, boundary, threshold, sqrt_2_ln_n, soft ? 1 : 2, y_dev,
                               thresholds_out_dev, static_cast<hipStream_t>(stream));
    }
    const size_t need = round256(vw_swt_denoise_workspace_bytes(b, n, levels));
    if (need > kScratchBytes)
        return set_err(ctx, VW_E_WORKSPACE,
                       "batch of %lld x %lld samples at %d levels needs %zu bytes of scratch; the allocation-free entry point "
                       "holds %zu (VW_NOALLOC_SCRATCH_BYTES): split the batch or use vw_swt_denoise_batch_dev",
                       static_cast<long long>(b), static_cast<long long>(n), levels, need, kScratchBytes);
    if (ctx->scratch == nullptr)
        return set_err(ctx, VW_E_WORKSPACE,
                       "the static scratch array of the allocation-free entry point could not be located on device %d "
                       "(hipGetSymbolAddress failed in vw_init): use vw_swt_denoise_batch_dev", ctx->device);
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    char* base = static_cast<char*>(ctx->scratch);  // device address of this GPU's array, never the symbol itself
    // what the coefficient block leaves of the array serves multi-pass plans as their intermediate plane
    ctx->ws_borrow.p = base + need;
    ctx->ws_borrow.bytes = kScratchBytes - need;
    rc = swt_denoise_run(ctx, x_dev, b, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold, sqrt_2_ln_n, soft, y_dev,
                         thresholds_out_dev, base, static_cast<hipStream_t>(stream));
    ctx->ws_borrow.p = nullptr;
    ctx->ws_borrow.bytes = 0;
    return rc;
}

// forward -> per-row threshold -> thresholding inverse, all in the workspace `wsb` (coefficient block + thresholds)
static int swt_denoise_run(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int32_t levels, const double* h, int32_t lh,
                           const double* g, int32_t lg, const double* hr, int32_t lhr, const double* gr, int32_t lgr,
                           int32_t boundary, double threshold, double sqrt_2_ln_n, int32_t soft, double* y_dev,
                           double* thresholds_out_dev, char* wsb, hipStream_t s) {
    int rc;
    double* coeffs = reinterpret_cast<double*>(wsb);
    double* thr = reinterpret_cast<double*>(wsb + static_cast<size_t>(b) * (levels + 1) * n * 8);

    rc = forward_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, 0, coeffs, nullptr, s);
    if (rc != VW_OK) return rc;
    if (threshold < 0) {
        rc = universal_threshold_dev(ctx, coeffs, b, n, levels, sqrt_2_ln_n, thr, s);
        if (rc != VW_OK) return rc;
    } else {
        hipLaunchKernelGGL(k_fill, dim3(static_cast<unsigned>((b + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, thr, b,
                           threshold);
        VW_HIP(ctx, hipGetLastError());
    }
    if (thresholds_out_dev != nullptr)
        VW_HIP(ctx, hipMemcpyAsync(thresholds_out_dev, thr, static_cast<size_t>(b) * 8, hipMemcpyDeviceToDevice, s));
    const int mode = soft ? 1 : 2;
    // fused inverse thresholds the detail bands as they are loaded; the generic
    // path (SYMMETRIC levels outside the tile kernel, truncated filters) thresholds the block in place first
    bool generic = (boundary == VW_BOUNDARY_SYMMETRIC) && !sym_inverse_all_tiled(ctx, lhr, lgr, levels, n);
    for (int j = 1; j <= levels; ++j) {
        const int64_t st = static_cast<int64_t>(1) << (j - 1);
        if (truncated_taps(lhr, st, n) != lhr || truncated_taps(lgr, st, n) != lgr) generic = true;
    }
    if (generic) {
        const int64_t cnt = static_cast<int64_t>(levels) * n;
        int64_t bx = (cnt + kThreads - 1) / kThreads;
        if (bx > 1024) bx = 1024;
        for (int64_t r0 = 0; r0 < b; r0 += 65535) {
            const int64_t nr = (b - r0) < 65535 ? (b - r0) : 65535;
            hipLaunchKernelGGL(k_threshold_rows, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(nr)), dim3(kThreads), 0, s,
                               coeffs + r0 * (levels + 1) * n, n, levels, thr + r0, static_cast<int64_t>(1), static_cast<int64_t>(0), mode);
        }
        VW_HIP(ctx, hipGetLastError());
        return inverse_dev(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, 0, 0xffffffffu, 0, nullptr, y_dev, nullptr, s);
    }
    return inverse_dev(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, 0, 0xffffffffu, mode, thr, y_dev, nullptr, s);
}

extern "C" int32_t vw_swt_denoise_batch(vw_ctx* ctx, const double* x, int64_t b, int64_t n, int32_t levels,
                                        const double* h, int32_t lh, const double* g, int32_t lg, const double* hr,
                                        int32_t lhr, const double* gr, int32_t lgr, int32_t boundary, double threshold,
                                        double sqrt_2_ln_n, int32_t soft, double* y, double* thresholds_out) {
    int rc = forward_checks(ctx, x, b, n, levels, h, lh, g, lg, boundary, 0, y);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x, b, n, levels, hr, lhr, gr, lgr, boundary, 0, y);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t rows = stage_rows(b, n, levels + 3);
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(rows) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, static_cast<size_t>(rows) * n * 8 + static_cast<size_t>(rows) * 8);
        if (rc != VW_OK) return rc;
    }
    double* dx = static_cast<double*>(ctx->stage_in.p);
    double* dy = static_cast<double*>(ctx->stage_out.p);
    double* dthr = dy + rows * n;
    for (int64_t r0 = 0; r0 < b; r0 += rows) {
        const int64_t nr = (b - r0) < rows ? (b - r0) : rows;
        VW_HIP(ctx, hipMemcpyAsync(dx, x + r0 * n, static_cast<size_t>(nr) * n * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = vw_swt_denoise_batch_dev(ctx, dx, nr, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, threshold,
                                      sqrt_2_ln_n, soft, dy, thresholds_out != nullptr ? dthr : nullptr, nullptr, 0, ctx->stream);
        if (rc != VW_OK) return rc;
        VW_HIP(ctx, hipMemcpyAsync(y + r0 * n, dy, static_cast<size_t>(nr) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (thresholds_out != nullptr)
            VW_HIP(ctx, hipMemcpyAsync(thresholds_out + r0, dthr, static_cast<size_t>(nr) * 8, hipMemcpyDeviceToHost, ctx->stream));
        VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VW_OK;
}

// ---------------------------------------------------------------------------
// WaveletDenoiser
// ---------------------------------------------------------------------------

extern "C" size_t vw_wavelet_denoise_workspace_bytes(int64_t b, int64_t n, int32_t levels) {
    if (b <= 0 || n <= 0 || levels < 1) return 0;
    // coefficient block + per-row sigma + per-(row, level) thresholds
    return round256(static_cast<size_t>(b) * static_cast<size_t>(levels + 1) * static_cast<size_t>(n) * 8) +
           round256(static_cast<size_t>(b) * 8) + round256(static_cast<size_t>(b) * static_cast<size_t>(levels) * 8);
}

// sigma per row (MAD of band 0), then one workgroup per (row, level) selects the threshold
static int denoiser_thresholds_dev(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int levels, int method,
                                   int level_scaled, double log_n, double* sigma, double* thr, hipStream_t s) {
    if (method < VW_THRESHOLD_UNIVERSAL || method > VW_THRESHOLD_BAYES)
        return set_err(ctx, VW_E_UNSUPPORTED, method == VW_THRESHOLD_FIXED
                                                  ? "Fixed threshold method requires explicit threshold value"
                                                  : "Unknown threshold selection method");
    // sigma = median(|d_1|) / 0.6745: the universal-threshold kernel with a factor of exactly 1.0
    int rc = universal_threshold_dev(ctx, coeffs, b, n, levels, 1.0, sigma, s);
    if (rc != VW_OK) return rc;
    DenoiserThrArgs a;
    memset(&a, 0, sizeof(a));
    a.coeffs = coeffs;
    a.row_stride = static_cast<int64_t>(levels + 1) * n;
    a.band_stride = n;
    a.n = n;
    a.nrows = b;
    a.levels = levels;
    a.method = method;
    a.level_scaled = level_scaled;
    a.log_n = log_n < 0 ? log(static_cast<double>(n)) : log_n;
    a.sigma = sigma;
    a.thr = thr;
    const int64_t nblocks = b * levels;
    if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    if (method == VW_THRESHOLD_BAYES)
        hipLaunchKernelGGL(k_denoiser_bayes, dim3(static_cast<unsigned>((nblocks + kChainRows - 1) / kChainRows)), dim3(64), 0, s, a);
    else if (method == VW_THRESHOLD_UNIVERSAL || method == VW_THRESHOLD_MINIMAX)
        hipLaunchKernelGGL(k_denoiser_sigma_rules, dim3(static_cast<unsigned>((nblocks + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL(k_denoiser_thresholds, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, s, a);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_denoiser_thresholds_batch_dev(vw_ctx* ctx, const double* coeffs_dev, int64_t b, int64_t n,
                                                    int32_t levels, int32_t method, int32_t level_scaled, double log_n,
                                                    double* thresholds_dev, double* sigma_scratch_dev, void* stream) {
    if (ctx == nullptr) return VW_E_NULL;
    if (b < 0 || n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (b > 0 && (coeffs_dev == nullptr || thresholds_dev == nullptr || sigma_scratch_dev == nullptr))
        return set_err(ctx, VW_E_NULL, "Result cannot be null");
    if (b > 0 && n == 0) return set_err(ctx, VW_E_EMPTY, "Signal cannot be empty");
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of levels %d", levels);
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    return denoiser_thresholds_dev(ctx, coeffs_dev, b, n, levels, method, level_scaled, log_n, sigma_scratch_dev,
                                   thresholds_dev, static_cast<hipStream_t>(stream));
}

extern "C" int32_t vw_denoiser_thresholds_batch(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int32_t levels,
                                                int32_t method, int32_t level_scaled, double log_n, double* thresholds) {
    if (ctx == nullptr) return VW_E_NULL;
    if (b < 0 || n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (b > 0 && (coeffs == nullptr || thresholds == nullptr)) return set_err(ctx, VW_E_NULL, "Result cannot be null");
    if (b > 0 && n == 0) return set_err(ctx, VW_E_EMPTY, "Signal cannot be empty");
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of levels %d", levels);
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cbytes = static_cast<size_t>(b) * (levels + 1) * n * 8;
    int rc = ensure(ctx, ctx->stage_in, cbytes);
    if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, round256(static_cast<size_t>(b) * 8) + static_cast<size_t>(b) * levels * 8);
    if (rc != VW_OK) return rc;
    double* dc = static_cast<double*>(ctx->stage_in.p);
    double* dsig = static_cast<double*>(ctx->stage_out.p);
    double* dthr = reinterpret_cast<double*>(static_cast<char*>(ctx->stage_out.p) + round256(static_cast<size_t>(b) * 8));
    VW_HIP(ctx, hipMemcpyAsync(dc, coeffs, cbytes, hipMemcpyHostToDevice, ctx->stream));
    rc = denoiser_thresholds_dev(ctx, dc, b, n, levels, method, level_scaled, log_n, dsig, dthr, ctx->stream);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(thresholds, dthr, static_cast<size_t>(b) * levels * 8, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

// thresholds [b][levels] applied to the detail bands of coeffs, then the inverse: on load inside the inverse
// kernels where they handle the configuration, else in place ahead of the generic per-level inverse
static int threshold_inverse_dev(vw_ctx* ctx, double* coeffs, int64_t b, int64_t n, int levels, const double* hr, int lhr,
                                 const double* gr, int lgr, int boundary, uint32_t tflags, int soft, const double* thr,
                                 double* y_dev, hipStream_t s) {
    const bool single = (tflags & VW_FLAG_SINGLE_LEVEL) != 0;
    const int mode = soft ? 3 : 2;  // WaveletDenoiser.applyThreshold: the `abs <= thr ? 0.0 : ...` form
    bool generic = (boundary == VW_BOUNDARY_SYMMETRIC) && !sym_inverse_all_tiled(ctx, lhr, lgr, levels, n);
    if (!single) {
        for (int j = 1; j <= levels; ++j) {
            const int64_t st = static_cast<int64_t>(1) << (j - 1);
            if (truncated_taps(lhr, st, n) != lhr || truncated_taps(lgr, st, n) != lgr) generic = true;
        }
    }
    if (generic) {
        const int64_t cnt = static_cast<int64_t>(levels) * n;
        int64_t bx = (cnt + kThreads - 1) / kThreads;
        if (bx > 1024) bx = 1024;
        for (int64_t r0 = 0; r0 < b; r0 += 65535) {
            const int64_t nr = (b - r0) < 65535 ? (b - r0) : 65535;
            hipLaunchKernelGGL(k_threshold_rows, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(nr)), dim3(kThreads), 0, s,
                               coeffs + r0 * (levels + 1) * n, n, levels, thr + r0 * levels, static_cast<int64_t>(levels),
                               static_cast<int64_t>(1), mode);
        }
        VW_HIP(ctx, hipGetLastError());
        return inverse_dev(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, tflags, 0xffffffffu, 0, nullptr, y_dev,
                           nullptr, s);
    }
    return inverse_dev(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, tflags, 0xffffffffu, mode, thr, y_dev, nullptr, s,
                       levels, 1);
}

extern "C" int32_t vw_wavelet_denoise_batch_dev(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n, int32_t levels,
                                                const double* h, int32_t lh, const double* g, int32_t lg,
                                                const double* hr, int32_t lhr, const double* gr, int32_t lgr,
                                                int32_t boundary, uint32_t flags, int32_t method, int32_t soft,
                                                double fixed_threshold, double log_n, double* y_dev,
                                                double* thresholds_out_dev, void* workspace_dev, size_t workspace_bytes,
                                                void* stream) {
    const uint32_t tflags = flags & (VW_FLAG_SINGLE_LEVEL | VW_FLAG_FMA);
    const bool single = (tflags & VW_FLAG_SINGLE_LEVEL) != 0;
    int rc = forward_checks(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, tflags, y_dev);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x_dev, b, n, levels, hr, lhr, gr, lgr, boundary, tflags, y_dev);
    if (rc != VW_OK) return rc;
    if (method < VW_THRESHOLD_UNIVERSAL || method > VW_THRESHOLD_FIXED)
        return set_err(ctx, VW_E_UNSUPPORTED, "Unknown threshold selection method");
    if (method == VW_THRESHOLD_FIXED && !single)
        return set_err(ctx, VW_E_UNSUPPORTED, "Fixed threshold method requires explicit threshold value");
    if (b == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a workspace the caller passes must be large enough whichever kernels the plan takes (the short-row kernels below
    // would not touch it): the error contract does not depend on the row length
    const size_t need = vw_wavelet_denoise_workspace_bytes(b, n, levels);
    if (workspace_dev != nullptr && workspace_bytes < need)
        return set_err(ctx, VW_E_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    // short rows, rules that are functions of sigma alone (or a fixed threshold): the single-launch row kernels, in the
    // reference's arithmetic only
    if (!(flags & VW_FLAG_FMA) && row_denoise_eligible(ctx, n, levels, lh, lg, lhr, lgr, boundary) &&
        (method == VW_THRESHOLD_UNIVERSAL || method == VW_THRESHOLD_MINIMAX ||
         (method == VW_THRESHOLD_FIXED && !(fixed_threshold < 0)))) {
        RowRule rr;
        const bool fixed = method == VW_THRESHOLD_FIXED;
        rr.rule = fixed ? 0 : 1;
        rr.method = method;
        rr.level_scaled = single ? 0 : 1;
        rr.log_n = log_n < 0 ? log(static_cast<double>(n)) : log_n;
        // a fixed threshold applies to the single level the fixed form has; thresholds_out is [b][levels] either way
        return row_denoise_dev(ctx, x_dev, b, n, levels, h, lh, g, hr, gr, boundary, fixed ? fixed_threshold : -1.0, 0.0,
                               soft ? 3 : 2, y_dev, thresholds_out_dev, s, rr);
    }
    char* wsb;
    if (workspace_dev != nullptr) {
        wsb = static_cast<char*>(workspace_dev);
    } else {
        rc = ensure(ctx, ctx->denoise_ws, need);
        if (rc != VW_OK) return rc;
        wsb = static_cast<char*>(ctx->denoise_ws.p);
    }
    const size_t cbytes = round256(static_cast<size_t>(b) * (levels + 1) * n * 8);
    double* coeffs = reinterpret_cast<double*>(wsb);
    double* sigma = reinterpret_cast<double*>(wsb + cbytes);
    double* thr = reinterpret_cast<double*>(wsb + cbytes + round256(static_cast<size_t>(b) * 8));

    rc = forward_dev(ctx, x_dev, b, n, levels, h, lh, g, lg, boundary, tflags, coeffs, nullptr, s);
    if (rc != VW_OK) return rc;
    if (method == VW_THRESHOLD_FIXED) {
        hipLaunchKernelGGL(k_fill, dim3(static_cast<unsigned>((b + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, thr, b,
                           fixed_threshold);
        VW_HIP(ctx, hipGetLastError());
    } else {
        rc = denoiser_thresholds_dev(ctx, coeffs, b, n, levels, method, single ? 0 : 1, log_n, sigma, thr, s);
        if (rc != VW_OK) return rc;
    }
    if (thresholds_out_dev != nullptr)
        VW_HIP(ctx, hipMemcpyAsync(thresholds_out_dev, thr, static_cast<size_t>(b) * levels * 8, hipMemcpyDeviceToDevice, s));
    return threshold_inverse_dev(ctx, coeffs, b, n, levels, hr, lhr, gr, lgr, boundary, tflags, soft, thr, y_dev, s);
}

extern "C" int32_t vw_wavelet_denoise_batch(vw_ctx* ctx, const double* x, int64_t b, int64_t n, int32_t levels,
                                            const double* h, int32_t lh, const double* g, int32_t lg, const double* hr,
                                            int32_t lhr, const double* gr, int32_t lgr, int32_t boundary, uint32_t flags,
                                            int32_t method, int32_t soft, double fixed_threshold, double log_n, double* y,
                                            double* thresholds_out) {
    const uint32_t tflags = flags & (VW_FLAG_SINGLE_LEVEL | VW_FLAG_FMA);
    int rc = forward_checks(ctx, x, b, n, levels, h, lh, g, lg, boundary, tflags, y);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x, b, n, levels, hr, lhr, gr, lgr, boundary, tflags, y);
    if (rc != VW_OK) return rc;
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t rows = stage_rows(b, n, levels + 3);
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(rows) * n * 8);
    if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, round256(static_cast<size_t>(rows) * n * 8) + static_cast<size_t>(rows) * levels * 8);
    if (rc != VW_OK) return rc;
    double* dx = static_cast<double*>(ctx->stage_in.p);
    double* dy = static_cast<double*>(ctx->stage_out.p);
    double* dthr = reinterpret_cast<double*>(static_cast<char*>(ctx->stage_out.p) + round256(static_cast<size_t>(rows) * n * 8));
    for (int64_t r0 = 0; r0 < b; r0 += rows) {
        const int64_t nr = (b - r0) < rows ? (b - r0) : rows;
        VW_HIP(ctx, hipMemcpyAsync(dx, x + r0 * n, static_cast<size_t>(nr) * n * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = vw_wavelet_denoise_batch_dev(ctx, dx, nr, n, levels, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft,
                                          fixed_threshold, log_n, dy, thresholds_out != nullptr ? dthr : nullptr, nullptr, 0,
                                          ctx->stream);
        if (rc != VW_OK) return rc;
        VW_HIP(ctx, hipMemcpyAsync(y + r0 * n, dy, static_cast<size_t>(nr) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (thresholds_out != nullptr)
            VW_HIP(ctx, hipMemcpyAsync(thresholds_out + r0 * levels, dthr, static_cast<size_t>(nr) * levels * 8,
                                       hipMemcpyDeviceToHost, ctx->stream));
        VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VW_OK;
}

// ---------------------------------------------------------------------------
// MODWTStreamingDenoiser: lock-step streams with their noise windows in device memory
// ---------------------------------------------------------------------------

struct vw_stream_state {
    int device;
    int64_t streams, window;
    double* win;     // [streams][window]
    double* level;   // [streams]
    int64_t index;   // ring position, common to the streams
    int64_t samples; // per stream
};

extern "C" int32_t vw_stream_state_create(vw_ctx* ctx, int64_t streams, int64_t window_size, vw_stream_state** out) {
    if (ctx == nullptr) return VW_E_NULL;
    if (out == nullptr) return set_err(ctx, VW_E_NULL, "state pointer cannot be null");
    *out = nullptr;
    if (streams <= 0) return set_err(ctx, VW_E_NOT_POSITIVE, "Stream count must be positive");
    if (window_size <= 0) return set_err(ctx, VW_E_NOT_POSITIVE, "Noise window size must be positive");
    if (window_size > 0x7fffffffLL || streams > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "state too large");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    vw_stream_state* st = new (std::nothrow) vw_stream_state();
    if (st == nullptr) return set_err(ctx, VW_E_NOMEM, "out of host memory");
    st->device = ctx->device;
    st->streams = streams;
    st->window = window_size;
    st->win = nullptr;
    st->level = nullptr;
    st->index = 0;
    st->samples = 0;
    const size_t wbytes = static_cast<size_t>(streams) * static_cast<size_t>(window_size) * 8;
    if (hipMalloc(reinterpret_cast<void**>(&st->win), wbytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&st->level), static_cast<size_t>(streams) * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (st->win != nullptr) (void)hipFree(st->win);
        delete st;
        return set_err(ctx, VW_E_NOMEM, "cannot allocate %zu bytes of noise windows", wbytes);
    }
    hipError_t e = hipMemsetAsync(st->win, 0, wbytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(st->level, 0, static_cast<size_t>(streams) * 8, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(st->win);
        (void)hipFree(st->level);
        delete st;
        return set_err(ctx, VW_E_HIP, "HIP error: %s", hipGetErrorString(e));
    }
    *out = st;
    return VW_OK;
}

extern "C" int32_t vw_stream_state_destroy(vw_ctx* ctx, vw_stream_state* state) {
    if (ctx == nullptr) return VW_E_NULL;
    if (state == nullptr) return VW_OK;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(state->device));
    (void)hipDeviceSynchronize();
    (void)hipFree(state->win);
    (void)hipFree(state->level);
    delete state;
    return VW_OK;
}

extern "C" int32_t vw_stream_state_noise_levels(vw_ctx* ctx, vw_stream_state* state, double* levels) {
    if (ctx == nullptr) return VW_E_NULL;
    if (state == nullptr || levels == nullptr) return set_err(ctx, VW_E_NULL, "state and levels cannot be null");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(state->device));
    VW_HIP(ctx, hipDeviceSynchronize());
    VW_HIP(ctx, hipMemcpy(levels, state->level, static_cast<size_t>(state->streams) * 8, hipMemcpyDeviceToHost));
    return VW_OK;
}

extern "C" int32_t vw_stream_state_window(vw_ctx* ctx, vw_stream_state* state, double* window, int64_t* window_index) {
    if (ctx == nullptr) return VW_E_NULL;
    if (state == nullptr || window == nullptr) return set_err(ctx, VW_E_NULL, "state and window cannot be null");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(state->device));
    VW_HIP(ctx, hipDeviceSynchronize());
    VW_HIP(ctx, hipMemcpy(window, state->win, static_cast<size_t>(state->streams) * static_cast<size_t>(state->window) * 8,
                          hipMemcpyDeviceToHost));
    if (window_index != nullptr) *window_index = state->index;
    return VW_OK;
}

extern "C" int64_t vw_stream_state_samples_processed(const vw_stream_state* state) {
    return state != nullptr ? state->samples : 0;
}

extern "C" int32_t vw_streaming_denoise_batch_dev(vw_ctx* ctx, vw_stream_state* state, const double* x_dev, int64_t b,
                                                  int64_t n, const double* h, int32_t lh, const double* g, int32_t lg,
                                                  const double* hr, int32_t lhr, const double* gr, int32_t lgr,
                                                  int32_t boundary, uint32_t flags, int32_t method, int32_t soft,
                                                  double threshold_multiplier, int32_t noise_estimation, double log_n,
                                                  double sqrt_2_ln_n, double* y_dev, double* thresholds_out_dev,
                                                  void* workspace_dev, size_t workspace_bytes, void* stream) {
    const uint32_t tflags = (flags & VW_FLAG_FMA) | VW_FLAG_SINGLE_LEVEL;
    if (ctx == nullptr) return VW_E_NULL;
    if (state == nullptr) return set_err(ctx, VW_E_NULL, "Stream state cannot be null");
    if (b > 0 && n == 0) return set_err(ctx, VW_E_EMPTY, "Samples cannot be null or empty");
    int rc = forward_checks(ctx, x_dev, b, n, 1, h, lh, g, lg, boundary, tflags, y_dev);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x_dev, b, n, 1, hr, lhr, gr, lgr, boundary, tflags, y_dev);
    if (rc != VW_OK) return rc;
    if (b != state->streams)
        return set_err(ctx, VW_E_LENGTH_MISMATCH, "block of %lld rows given to a state of %lld streams", static_cast<long long>(b),
                       static_cast<long long>(state->streams));
    if (noise_estimation < VW_NOISE_MAD || noise_estimation > VW_NOISE_FIXED)
        return set_err(ctx, VW_E_UNSUPPORTED, "Unknown noise estimation method");
    if (!(threshold_multiplier > 0)) return set_err(ctx, VW_E_NOT_POSITIVE, "Threshold multiplier must be positive");
    if (n > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "blocks of more than 2^31-1 samples are not supported");
    if (state->device != ctx->device) return set_err(ctx, VW_E_UNSUPPORTED, "stream state belongs to another device");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t need = vw_wavelet_denoise_workspace_bytes(b, n, 1);
    char* wsb;
    if (workspace_dev != nullptr) {
        if (workspace_bytes < need) return set_err(ctx, VW_E_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        wsb = static_cast<char*>(workspace_dev);
    } else {
        rc = ensure(ctx, ctx->denoise_ws, need);
        if (rc != VW_OK) return rc;
        wsb = static_cast<char*>(ctx->denoise_ws.p);
    }
    const size_t cbytes = round256(static_cast<size_t>(b) * 2 * n * 8);
    double* coeffs = reinterpret_cast<double*>(wsb);
    double* sigma = reinterpret_cast<double*>(wsb + cbytes);
    double* thr = reinterpret_cast<double*>(wsb + cbytes + round256(static_cast<size_t>(b) * 8));

    rc = forward_dev(ctx, x_dev, b, n, 1, h, lh, g, lg, boundary, tflags, coeffs, nullptr, s);
    if (rc != VW_OK) return rc;
    const bool plain = fabs(threshold_multiplier - 1.0) < 1e-10;
    if (noise_estimation != VW_NOISE_FIXED || !plain) {
        StreamNoiseArgs a;
        memset(&a, 0, sizeof(a));
        a.details = coeffs;  // band 0 of a single-level block is the detail band
        a.row_stride = 2 * n;
        a.n = n;
        a.nrows = b;
        a.window = state->win;
        a.w = state->window;
        a.index0 = state->index;
        a.estimation = noise_estimation;
        a.method = method;
        a.multiplier = threshold_multiplier;
        a.log_n = log_n < 0 ? log(static_cast<double>(n)) : log_n;
        a.sqrt_2_ln_n = sqrt_2_ln_n < 0 ? sqrt(2.0 * a.log_n) : sqrt_2_ln_n;
        a.level = state->level;
        a.thr = thr;
        a.lds_cap = 6144;
        if (b > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
        // one launch updates the ring, estimates the level and (with a multiplier) derives the threshold; the STD
        // estimator runs between two such launches in a kernel of its own that walks 16 rings per wavefront
        auto launch = [&](int do_update, int do_level, int want_thr) {
            a.do_update = do_update;
            a.do_level = do_level;
            a.want_thr = want_thr;
            int64_t staged = 0;  // doubles of LDS the 256-lane kernel stages a ring or a block in
            if (do_level && state->window <= a.lds_cap) staged = state->window;
            if (want_thr && n <= a.lds_cap && n > staged) staged = n;
            const size_t lds = staged > 0 ? static_cast<size_t>((staged + 1) & ~static_cast<int64_t>(1)) * 8 : 16;
            const dim3 grid(static_cast<unsigned>(b));
            if (do_level == do_update && !(do_level && noise_estimation == VW_NOISE_STD)) {  // MAD, or the threshold alone
                const int kw = do_update ? reg_kpl(ctx->opt, state->window) : 1, kn = want_thr ? reg_kpl(ctx->opt, n) : 1;
                if (kw != 0 && kn != 0) {
                    if (do_update) switch (kw) {
                        case 4: hipLaunchKernelGGL(k_stream_ring_mad_reg<4>, grid, dim3(64), 0, s, a); break;
                        case 8: hipLaunchKernelGGL(k_stream_ring_mad_reg<8>, grid, dim3(64), 0, s, a); break;
                        case 16: hipLaunchKernelGGL(k_stream_ring_mad_reg<16>, grid, dim3(64), 0, s, a); break;
                        default: hipLaunchKernelGGL(k_stream_ring_mad_reg<32>, grid, dim3(64), 0, s, a); break;
                    }
                    if (want_thr) switch (kn) {
                        case 4: hipLaunchKernelGGL(k_stream_threshold_reg<4>, grid, dim3(64), 0, s, a); break;
                        case 8: hipLaunchKernelGGL(k_stream_threshold_reg<8>, grid, dim3(64), 0, s, a); break;
                        case 16: hipLaunchKernelGGL(k_stream_threshold_reg<16>, grid, dim3(64), 0, s, a); break;
                        default: hipLaunchKernelGGL(k_stream_threshold_reg<32>, grid, dim3(64), 0, s, a); break;
                    }
                    return;
                }
            }
            hipLaunchKernelGGL(k_stream_noise<kThreads>, dim3(static_cast<unsigned>(b)), dim3(kThreads), lds, s, a);
        };
        if (noise_estimation == VW_NOISE_STD && !ctx->opt.no_std_chains) {
            launch(1, 0, 0);
            hipLaunchKernelGGL(k_stream_std, dim3(static_cast<unsigned>((b + kChainRows - 1) / kChainRows)), dim3(64), 0, s, a);
            if (!plain) launch(0, 0, 1);
        } else if (noise_estimation != VW_NOISE_FIXED) {
            launch(1, 1, plain ? 0 : 1);
        } else {
            launch(0, 0, 1);
        }
        VW_HIP(ctx, hipGetLastError());
        if (noise_estimation != VW_NOISE_FIXED) state->index = (state->index + (n <= state->window ? n : state->window)) % state->window;
    }
    if (plain) {
        // WaveletDenoiser.denoise(samples, method, type): FIXED has no automatic rule
        if (method < VW_THRESHOLD_UNIVERSAL || method > VW_THRESHOLD_BAYES)
            return set_err(ctx, VW_E_UNSUPPORTED, method == VW_THRESHOLD_FIXED
                                                      ? "Fixed threshold method requires explicit threshold value"
                                                      : "Unknown threshold selection method");
        rc = denoiser_thresholds_dev(ctx, coeffs, b, n, 1, method, 0, log_n, sigma, thr, s);
        if (rc != VW_OK) return rc;
    } else if (method == VW_THRESHOLD_BAYES || method < VW_THRESHOLD_UNIVERSAL || method > VW_THRESHOLD_FIXED) {
        return set_err(ctx, VW_E_UNSUPPORTED, "Unknown threshold method: %d", method);
    }
    if (thresholds_out_dev != nullptr)
        VW_HIP(ctx, hipMemcpyAsync(thresholds_out_dev, thr, static_cast<size_t>(b) * 8, hipMemcpyDeviceToDevice, s));
    rc = threshold_inverse_dev(ctx, coeffs, b, n, 1, hr, lhr, gr, lgr, boundary, tflags, soft, thr, y_dev, s);
    if (rc == VW_OK) state->samples += n;
    return rc;
}

extern "C" int32_t vw_streaming_denoise_batch(vw_ctx* ctx, vw_stream_state* state, const double* x, int64_t b, int64_t n,
                                              const double* h, int32_t lh, const double* g, int32_t lg, const double* hr,
                                              int32_t lhr, const double* gr, int32_t lgr, int32_t boundary, uint32_t flags,
                                              int32_t method, int32_t soft, double threshold_multiplier,
                                              int32_t noise_estimation, double log_n, double sqrt_2_ln_n, double* y,
                                              double* thresholds_out) {
    const uint32_t tflags = (flags & VW_FLAG_FMA) | VW_FLAG_SINGLE_LEVEL;
    if (ctx == nullptr) return VW_E_NULL;
    if (state == nullptr) return set_err(ctx, VW_E_NULL, "Stream state cannot be null");
    if (b > 0 && n == 0) return set_err(ctx, VW_E_EMPTY, "Samples cannot be null or empty");
    int rc = forward_checks(ctx, x, b, n, 1, h, lh, g, lg, boundary, tflags, y);
    if (rc != VW_OK) return rc;
    rc = inverse_checks(ctx, x, b, n, 1, hr, lhr, gr, lgr, boundary, tflags, y);
    if (rc != VW_OK) return rc;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    VW_HIP(ctx, hipSetDevice(ctx->device));
    // the streams advance together, so the whole block is staged at once
    rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(b) * n * 8);
    if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, round256(static_cast<size_t>(b) * n * 8) + static_cast<size_t>(b) * 8);
    if (rc != VW_OK) return rc;
    double* dx = static_cast<double*>(ctx->stage_in.p);
    double* dy = static_cast<double*>(ctx->stage_out.p);
    double* dthr = reinterpret_cast<double*>(static_cast<char*>(ctx->stage_out.p) + round256(static_cast<size_t>(b) * n * 8));
    VW_HIP(ctx, hipMemcpyAsync(dx, x, static_cast<size_t>(b) * n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = vw_streaming_denoise_batch_dev(ctx, state, dx, b, n, h, lh, g, lg, hr, lhr, gr, lgr, boundary, flags, method, soft,
                                        threshold_multiplier, noise_estimation, log_n, sqrt_2_ln_n, dy,
                                        thresholds_out != nullptr ? dthr : nullptr, nullptr, 0, ctx->stream);
    if (rc != VW_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    VW_HIP(ctx, hipMemcpyAsync(y, dy, static_cast<size_t>(b) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (thresholds_out != nullptr)
        VW_HIP(ctx, hipMemcpyAsync(thresholds_out, dthr, static_cast<size_t>(b) * 8, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

// ---------------------------------------------------------------------------
// sparse detail bands (SWTResult.toSparse / SparseSWTResult.toFull)
// ---------------------------------------------------------------------------

static int sparse_checks(vw_ctx* ctx, const void* p0, int64_t b, int64_t n, int levels) {
    if (ctx == nullptr) return VW_E_NULL;
    if (b < 0 || n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (levels < 1 || levels > kMaxLevels) return set_err(ctx, VW_E_LEVEL, "Invalid number of levels %d", levels);
    if (n > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "indices are 32-bit: signals of more than 2^31-1 samples are not supported");
    if (b * levels > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    if (b > 0 && p0 == nullptr) return set_err(ctx, VW_E_NULL, "Details cannot be null");
    return VW_OK;
}

// offsets must be a prefix sum: starting at 0 or above, never decreasing, no band with more than n entries (the
// kernels write where it says, so a wrong array would send them outside the buffers sized from its last entry)
static int check_offsets(vw_ctx* ctx, const int64_t* offsets, int64_t bands, int64_t n) {
    if (offsets[0] < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative offset");
    for (int64_t k = 0; k < bands; ++k)
        if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] > n)
            return set_err(ctx, VW_E_LENGTH_MISMATCH, "offsets[%lld..%lld] is not the prefix sum of band counts", static_cast<long long>(k),
                           static_cast<long long>(k + 1));
    return VW_OK;
}

extern "C" int32_t vw_swt_sparsify_batch_dev(vw_ctx* ctx, const double* coeffs_dev, int64_t b, int64_t n, int32_t levels,
                                             double threshold, int64_t* counts_dev, const int64_t* offsets_dev,
                                             int32_t* indices_dev, double* values_dev, void* stream) {
    int rc = sparse_checks(ctx, coeffs_dev, b, n, levels);
    if (rc != VW_OK) return rc;
    if (b == 0 || n == 0) return VW_OK;
    if (indices_dev == nullptr && counts_dev == nullptr) return set_err(ctx, VW_E_NULL, "counts cannot be null in the counting pass");
    if (indices_dev != nullptr && (offsets_dev == nullptr || values_dev == nullptr))
        return set_err(ctx, VW_E_NULL, "offsets and values cannot be null in the fill pass");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    SparsifyArgs a;
    memset(&a, 0, sizeof(a));
    a.coeffs = coeffs_dev;
    a.row_stride = static_cast<int64_t>(levels + 1) * n;
    a.band_stride = n;
    a.n = n;
    a.levels = levels;
    a.threshold = threshold;
    a.counts = counts_dev;
    a.offsets = offsets_dev;
    a.indices = indices_dev;
    a.values = values_dev;
    hipLaunchKernelGGL(k_sparsify, dim3(static_cast<unsigned>(b * levels)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_swt_sparsify_batch(vw_ctx* ctx, const double* coeffs, int64_t b, int64_t n, int32_t levels, double threshold,
                                         int64_t* counts, const int64_t* offsets, int32_t* indices, double* values) {
    int rc = sparse_checks(ctx, coeffs, b, n, levels);
    if (rc != VW_OK) return rc;
    if (b == 0 || n == 0) return VW_OK;
    if (indices == nullptr && counts == nullptr) return set_err(ctx, VW_E_NULL, "counts cannot be null in the counting pass");
    if (indices != nullptr && (offsets == nullptr || values == nullptr))
        return set_err(ctx, VW_E_NULL, "offsets and values cannot be null in the fill pass");
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t bands = b * levels;
    const size_t cbytes = static_cast<size_t>(b) * (levels + 1) * n * 8;
    rc = ensure(ctx, ctx->stage_in, cbytes);
    if (rc != VW_OK) return rc;
    double* dc = static_cast<double*>(ctx->stage_in.p);
    VW_HIP(ctx, hipMemcpyAsync(dc, coeffs, cbytes, hipMemcpyHostToDevice, ctx->stream));
    if (indices == nullptr) {
        rc = ensure(ctx, ctx->stage_out, static_cast<size_t>(bands) * 8);
        if (rc != VW_OK) return rc;
        int64_t* dcnt = static_cast<int64_t*>(ctx->stage_out.p);
        rc = vw_swt_sparsify_batch_dev(ctx, dc, b, n, levels, threshold, dcnt, nullptr, nullptr, nullptr, ctx->stream);
        if (rc != VW_OK) return rc;
        VW_HIP(ctx, hipMemcpyAsync(counts, dcnt, static_cast<size_t>(bands) * 8, hipMemcpyDeviceToHost, ctx->stream));
        VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return VW_OK;
    }
    const int64_t total = offsets[bands];  // the caller's prefix sum ends with the number of entries
    rc = check_offsets(ctx, offsets, bands, n);
    if (rc != VW_OK) return rc;
    const size_t obytes = round256(static_cast<size_t>(bands + 1) * 8);
    const size_t cbytes2 = round256(static_cast<size_t>(bands) * 8);
    const size_t ibytes = round256(static_cast<size_t>(total) * 4);
    rc = ensure(ctx, ctx->stage_out, obytes + cbytes2 + ibytes + static_cast<size_t>(total) * 8 + 256);
    if (rc != VW_OK) return rc;
    char* base = static_cast<char*>(ctx->stage_out.p);
    int64_t* doff = reinterpret_cast<int64_t*>(base);
    int64_t* dfound = reinterpret_cast<int64_t*>(base + obytes);
    int32_t* didx = reinterpret_cast<int32_t*>(base + obytes + cbytes2);
    double* dval = reinterpret_cast<double*>(base + obytes + cbytes2 + ibytes);
    VW_HIP(ctx, hipMemcpyAsync(doff, offsets, static_cast<size_t>(bands + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = vw_swt_sparsify_batch_dev(ctx, dc, b, n, levels, threshold, dfound, doff, didx, dval, ctx->stream);
    if (rc != VW_OK) return rc;
    if (total > 0) {
        VW_HIP(ctx, hipMemcpyAsync(indices, didx, static_cast<size_t>(total) * 4, hipMemcpyDeviceToHost, ctx->stream));
        VW_HIP(ctx, hipMemcpyAsync(values, dval, static_cast<size_t>(total) * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    int64_t* found = static_cast<int64_t*>(malloc(static_cast<size_t>(bands) * 8));
    if (found == nullptr) return set_err(ctx, VW_E_NOMEM, "out of host memory");
    hipError_t he = hipMemcpyAsync(found, dfound, static_cast<size_t>(bands) * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    int64_t bad_band = -1;
    if (he == hipSuccess)
        for (int64_t k = 0; k < bands && bad_band < 0; ++k)
            if (found[k] != offsets[k + 1] - offsets[k]) bad_band = k;
    const int64_t bad_found = bad_band >= 0 ? found[bad_band] : 0;
    free(found);
    if (he != hipSuccess) return set_err(ctx, VW_E_HIP, "sparsify copy back: %s", hipGetErrorString(he));
    if (bad_band >= 0)
        return set_err(ctx, VW_E_LENGTH_MISMATCH,
                       "band %lld holds %lld entries above the threshold but offsets leave room for %lld: the fill pass must see the "
                       "coefficients and threshold of the counting pass (surplus entries were dropped, nothing was written out of bounds)",
                       static_cast<long long>(bad_band), static_cast<long long>(bad_found),
                       static_cast<long long>(offsets[bad_band + 1] - offsets[bad_band]));
    return VW_OK;
}

extern "C" int32_t vw_swt_densify_batch_dev(vw_ctx* ctx, const int64_t* offsets_dev, const int32_t* indices_dev,
                                            const double* values_dev, int64_t b, int64_t n, int32_t levels,
                                            double* coeffs_out_dev, void* stream) {
    int rc = sparse_checks(ctx, coeffs_out_dev, b, n, levels);
    if (rc != VW_OK) return rc;
    if (b == 0 || n == 0) return VW_OK;
    if (offsets_dev == nullptr) return set_err(ctx, VW_E_NULL, "offsets cannot be null");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rowc = static_cast<int64_t>(levels + 1) * n;
    // the detail bands of a row are contiguous: one 2-D memset clears them and leaves the approximation bands alone
    VW_HIP(ctx, hipMemset2DAsync(coeffs_out_dev, static_cast<size_t>(rowc) * 8, 0, static_cast<size_t>(levels) * n * 8,
                                 static_cast<size_t>(b), s));
    DensifyArgs a;
    memset(&a, 0, sizeof(a));
    a.offsets = offsets_dev;
    a.indices = indices_dev;
    a.values = values_dev;
    a.out = coeffs_out_dev;
    a.row_stride = rowc;
    a.band_stride = n;
    a.n = n;
    a.levels = levels;
    hipLaunchKernelGGL(k_densify, dim3(static_cast<unsigned>(b * levels)), dim3(kThreads), 0, s, a);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_swt_densify_batch(vw_ctx* ctx, const int64_t* offsets, const int32_t* indices, const double* values,
                                        int64_t b, int64_t n, int32_t levels, double* coeffs_out) {
    int rc = sparse_checks(ctx, coeffs_out, b, n, levels);
    if (rc != VW_OK) return rc;
    if (b == 0 || n == 0) return VW_OK;
    if (offsets == nullptr) return set_err(ctx, VW_E_NULL, "offsets cannot be null");
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t bands = b * levels;
    const int64_t total = offsets[bands];
    rc = check_offsets(ctx, offsets, bands, n);
    if (rc != VW_OK) return rc;
    if (total > 0 && (indices == nullptr || values == nullptr)) return set_err(ctx, VW_E_NULL, "indices and values cannot be null");
    const size_t cbytes = static_cast<size_t>(b) * (levels + 1) * n * 8;
    const size_t obytes = round256(static_cast<size_t>(bands + 1) * 8);
    const size_t ibytes = round256(static_cast<size_t>(total) * 4);
    rc = ensure(ctx, ctx->stage_out, cbytes);
    if (rc == VW_OK) rc = ensure(ctx, ctx->stage_in, obytes + ibytes + static_cast<size_t>(total) * 8 + 256);
    if (rc != VW_OK) return rc;
    char* base = static_cast<char*>(ctx->stage_in.p);
    int64_t* doff = reinterpret_cast<int64_t*>(base);
    int32_t* didx = reinterpret_cast<int32_t*>(base + obytes);
    double* dval = reinterpret_cast<double*>(base + obytes + ibytes);
    double* dout = static_cast<double*>(ctx->stage_out.p);
    VW_HIP(ctx, hipMemcpyAsync(doff, offsets, static_cast<size_t>(bands + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (total > 0) {
        VW_HIP(ctx, hipMemcpyAsync(didx, indices, static_cast<size_t>(total) * 4, hipMemcpyHostToDevice, ctx->stream));
        VW_HIP(ctx, hipMemcpyAsync(dval, values, static_cast<size_t>(total) * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = vw_swt_densify_batch_dev(ctx, doff, didx, dval, b, n, levels, dout, ctx->stream);
    if (rc != VW_OK) return rc;
    // only the detail bands come back: the caller's approximation bands stay as they are
    VW_HIP(ctx, hipMemcpy2DAsync(coeffs_out, static_cast<size_t>(levels + 1) * n * 8, dout, static_cast<size_t>(levels + 1) * n * 8,
                                 static_cast<size_t>(levels) * n * 8, static_cast<size_t>(b), hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

// ---------------------------------------------------------------------------
// structure-of-arrays batch + layout conversion
// ---------------------------------------------------------------------------

extern "C" int32_t vw_transpose_dev(vw_ctx* ctx, const double* in_dev, int64_t rows, int64_t cols, double* out_dev, void* stream) {
    if (ctx == nullptr) return VW_E_NULL;
    if (rows < 0 || cols < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (rows == 0 || cols == 0) return VW_OK;
    if (in_dev == nullptr || out_dev == nullptr) return set_err(ctx, VW_E_NULL, "array cannot be null");
    const int64_t nblocks = ((rows + 31) / 32) * ((cols + 31) / 32);
    if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_transpose, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), in_dev,
                       out_dev, rows, cols);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_transpose(vw_ctx* ctx, const double* in, int64_t rows, int64_t cols, double* out) {
    if (ctx == nullptr) return VW_E_NULL;
    if (rows < 0 || cols < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (rows == 0 || cols == 0) return VW_OK;
    if (in == nullptr || out == nullptr) return set_err(ctx, VW_E_NULL, "array cannot be null");
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(rows) * cols * 8;
    int rc = ensure(ctx, ctx->stage_in, bytes);
    if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, bytes);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(ctx->stage_in.p, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = vw_transpose_dev(ctx, static_cast<const double*>(ctx->stage_in.p), rows, cols, static_cast<double*>(ctx->stage_out.p), ctx->stream);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(out, ctx->stage_out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

// k_soa_modwt_run exists for the Haar constants, the four-tap chain and the general form with an even number of taps up
// to 20; anything else stays with k_soa_modwt
static bool soa_run_supported(int flavour, int lh) {
    if (flavour == 1) return true;
    if (flavour == 2) return lh == 4;
    return lh >= 2 && lh <= 20 && lh % 2 == 0;
}

template <int L, int FLAV, int T>
static hipError_t launch_soa_run_t(const SoaArgs& a, bool vec2, hipStream_t stream) {
    const int64_t per_block = static_cast<int64_t>(kThreads) * (vec2 ? 2 : 1);
    const int64_t bchunks = (a.batch + per_block - 1) / per_block;
    const int64_t ttiles = (a.n + T - 1) / T;
    if (bchunks > 0x7fffffffLL / ttiles) return hipErrorInvalidConfiguration;
    const unsigned nblocks = static_cast<unsigned>(bchunks * ttiles);
    if (vec2)
        hipLaunchKernelGGL((k_soa_modwt_run<L, FLAV, T, 2>), dim3(nblocks), dim3(kThreads), 0, stream, a);
    else
        hipLaunchKernelGGL((k_soa_modwt_run<L, FLAV, T, 1>), dim3(nblocks), dim3(kThreads), 0, stream, a);
    return hipGetLastError();
}

template <int L, int FLAV>
static hipError_t launch_soa_run_l(const SoaArgs& a, bool vec2, int run_len, hipStream_t stream) {
    // the window is T + L - 1 samples (times two signals) in registers: long filters take the shorter run
    if (run_len == 0) run_len = L <= 12 ? 16 : 8;
    if (run_len == 32 && L <= 12) return launch_soa_run_t<L, FLAV, 32>(a, vec2, stream);
    if (run_len >= 16 && L <= 12) return launch_soa_run_t<L, FLAV, 16>(a, vec2, stream);
    if (run_len >= 16) return launch_soa_run_t<L, FLAV, 16>(a, vec2, stream);
    return launch_soa_run_t<L, FLAV, 8>(a, vec2, stream);
}

static hipError_t launch_soa_run(const SoaArgs& a, bool vec2, int run_len, hipStream_t stream) {
    if (a.flavour == 1) return launch_soa_run_l<2, 1>(a, vec2, run_len, stream);
    if (a.flavour == 2) return launch_soa_run_l<4, 2>(a, vec2, run_len, stream);
    switch (a.flen) {
        case 2: return launch_soa_run_l<2, 0>(a, vec2, run_len, stream);
        case 4: return launch_soa_run_l<4, 0>(a, vec2, run_len, stream);
        case 6: return launch_soa_run_l<6, 0>(a, vec2, run_len, stream);
        case 8: return launch_soa_run_l<8, 0>(a, vec2, run_len, stream);
        case 10: return launch_soa_run_l<10, 0>(a, vec2, run_len, stream);
        case 12: return launch_soa_run_l<12, 0>(a, vec2, run_len, stream);
        case 14: return launch_soa_run_l<14, 0>(a, vec2, run_len, stream);
        case 16: return launch_soa_run_l<16, 0>(a, vec2, run_len, stream);
        case 18: return launch_soa_run_l<18, 0>(a, vec2, run_len, stream);
        case 20: return launch_soa_run_l<20, 0>(a, vec2, run_len, stream);
        default: return hipErrorInvalidValue;
    }
}

static int soa_checks(vw_ctx* ctx, const void* in, const void* a, const void* d, int64_t batch, int64_t n, const double* h, int lh,
                      const double* g, int lg) {
    if (ctx == nullptr) return VW_E_NULL;
    if (h == nullptr || g == nullptr) return set_err(ctx, VW_E_NULL, "filter cannot be null");
    if (batch < 0 || n < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative size");
    if (lh < 1 || lh > kMaxTaps) return set_err(ctx, VW_E_LENGTH_MISMATCH, "filter length %d outside [1, %d]", lh, kMaxTaps);
    // the reference loops l < lowPass.length over both filters
    if (lg < lh) return set_err(ctx, VW_E_LENGTH_MISMATCH, "high-pass filter (%d taps) shorter than low-pass (%d)", lg, lh);
    if (batch > 0 && n > 0 && (in == nullptr || a == nullptr || d == nullptr)) return set_err(ctx, VW_E_NULL, "array cannot be null");
    if (batch > 0 && n > 0 && batch > INT64_MAX / n / 8) return set_err(ctx, VW_E_TOO_LARGE, "batch too large");
    return VW_OK;
}

extern "C" int32_t vw_batch_modwt_soa_dev(vw_ctx* ctx, const double* soa_signals_dev, int64_t batch, int64_t n, const double* h,
                                          int32_t lh, const double* g, int32_t lg, int32_t is_haar, double* soa_approx_dev,
                                          double* soa_detail_dev, void* stream) {
    int rc = soa_checks(ctx, soa_signals_dev, soa_approx_dev, soa_detail_dev, batch, n, h, lh, g, lg);
    if (rc != VW_OK) return rc;
    if (batch == 0 || n == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    SoaArgs a;
    memset(&a, 0, sizeof(a));
    a.in = soa_signals_dev;
    a.approx = soa_approx_dev;
    a.detail = soa_detail_dev;
    a.batch = batch;
    a.n = n;
    a.flen = lh;
    a.flavour = is_haar ? 1 : (lh == 4 ? 2 : 0);
    scale_taps(h, lh, a.taps.lo);
    scale_taps(g, lh, a.taps.hi);
    if (!ctx->opt.no_soa_runs && soa_run_supported(a.flavour, lh)) {
        const bool vec2 = batch % 2 == 0 && aligned16(soa_signals_dev) && aligned16(soa_approx_dev) && aligned16(soa_detail_dev);
        const hipError_t er = launch_soa_run(a, vec2, ctx->opt.soa_run_len, static_cast<hipStream_t>(stream));
        if (er == hipErrorInvalidConfiguration) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
        if (er != hipSuccess) return set_err(ctx, VW_E_HIP, "structure-of-arrays kernel launch: %s", hipGetErrorString(er));
        return VW_OK;
    }
    const int64_t nblocks = (batch * n + kThreads - 1) / kThreads;
    if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    hipLaunchKernelGGL(k_soa_modwt, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_batch_modwt_soa(vw_ctx* ctx, const double* soa_signals, int64_t batch, int64_t n, const double* h, int32_t lh,
                                      const double* g, int32_t lg, int32_t is_haar, double* soa_approx, double* soa_detail) {
    int rc = soa_checks(ctx, soa_signals, soa_approx, soa_detail, batch, n, h, lh, g, lg);
    if (rc != VW_OK) return rc;
    if (batch == 0 || n == 0) return VW_OK;
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);
    VW_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = static_cast<size_t>(batch) * n * 8;
    rc = ensure(ctx, ctx->stage_in, bytes);
    if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, 2 * bytes);
    if (rc != VW_OK) return rc;
    double* dx = static_cast<double*>(ctx->stage_in.p);
    double* da = static_cast<double*>(ctx->stage_out.p);
    double* dd = da + static_cast<size_t>(batch) * n;
    VW_HIP(ctx, hipMemcpyAsync(dx, soa_signals, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = vw_batch_modwt_soa_dev(ctx, dx, batch, n, h, lh, g, lg, is_haar, da, dd, ctx->stream);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(soa_approx, da, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipMemcpyAsync(soa_detail, dd, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

// ---------------------------------------------------------------------------
// raw convolution + validation
// ---------------------------------------------------------------------------

extern "C" int32_t vw_modwt_convolve_batch_dev(vw_ctx* ctx, const double* x_dev, int64_t b, int64_t n,
                                               const double* filter, int32_t filter_len, int32_t level,
                                               int32_t boundary, double* out_dev, void* stream) {
    if (ctx == nullptr) return VW_E_NULL;
    if (filter == nullptr) return set_err(ctx, VW_E_NULL, "filter cannot be null");
    if (level < 1) return set_err(ctx, VW_E_LEVEL, "level must be >= 1");
    int rc = check_common(ctx, b, n, level, filter_len, filter_len, boundary);
    if (rc != VW_OK) return rc;
    if (b > 0 && (x_dev == nullptr || out_dev == nullptr)) return set_err(ctx, VW_E_NULL, "signal cannot be null");
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    LevelArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x_dev;
    a.in_stride = n;
    a.out_lo = out_dev;
    a.out_lo_stride = n;
    a.out_hi = nullptr;
    a.n = n;
    a.nrows = b;
    a.stride = static_cast<int64_t>(1) << (level - 1);
    a.llo = filter_len;
    a.lhi = 0;
    for (int i = 0; i < kMaxTaps + 2; ++i) {
        a.taps.lo[i] = i < filter_len ? filter[i] : 0.0;
        a.taps.hi[i] = 0.0;
    }
    const int64_t per_row = (n + kThreads - 1) / kThreads;
    const int64_t nblocks = per_row * b;
    if (nblocks > 0x7fffffffLL) return set_err(ctx, VW_E_TOO_LARGE, "grid too large");
    if (boundary == VW_BOUNDARY_PERIODIC)
        hipLaunchKernelGGL(k_level_fwd<kPeriodic>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, s, a);
    else if (boundary == VW_BOUNDARY_ZERO_PADDING)
        hipLaunchKernelGGL(k_level_fwd<kZeroPad>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL(k_level_fwd<kSymmetric>, dim3(static_cast<unsigned>(nblocks)), dim3(kThreads), 0, s, a);
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_modwt_convolve_batch(vw_ctx* ctx, const double* x, int64_t b, int64_t n, const double* filter,
                                           int32_t filter_len, int32_t level, int32_t boundary, double* out) {
    if (ctx == nullptr) return VW_E_NULL;
    if (filter == nullptr) return set_err(ctx, VW_E_NULL, "filter cannot be null");
    if (level < 1) return set_err(ctx, VW_E_LEVEL, "level must be >= 1");
    int rc = check_common(ctx, b, n, level, filter_len, filter_len, boundary);
    if (rc != VW_OK) return rc;
    if (b > 0 && (x == nullptr || out == nullptr)) return set_err(ctx, VW_E_NULL, "signal cannot be null");
    if (b == 0) return VW_OK;
    VW_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(b) * n * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->stage_out, static_cast<size_t>(b) * n * 8);
        if (rc != VW_OK) return rc;
    }
    double* dx = static_cast<double*>(ctx->stage_in.p);
    double* dy = static_cast<double*>(ctx->stage_out.p);
    VW_HIP(ctx, hipMemcpyAsync(dx, x, static_cast<size_t>(b) * n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = vw_modwt_convolve_batch_dev(ctx, dx, b, n, filter, filter_len, level, boundary, dy, ctx->stream);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(out, dy, static_cast<size_t>(b) * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VW_OK;
}

extern "C" int32_t vw_first_non_finite_dev(vw_ctx* ctx, const double* x_dev, int64_t count, int64_t* first_bad_dev,
                                           void* stream) {
    if (ctx == nullptr) return VW_E_NULL;
    if (first_bad_dev == nullptr) return set_err(ctx, VW_E_NULL, "first_bad cannot be null");
    if (count < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative count");
    if (count > 0 && x_dev == nullptr) return set_err(ctx, VW_E_NULL, "signal cannot be null");
    VW_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    VW_HIP(ctx, hipMemsetAsync(first_bad_dev, 0xff, 8, s));
    if (count == 0) return VW_OK;
    int64_t blocks = (count + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_first_non_finite, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, x_dev, count,
                       reinterpret_cast<unsigned long long*>(first_bad_dev));
    VW_HIP(ctx, hipGetLastError());
    return VW_OK;
}

extern "C" int32_t vw_first_non_finite(vw_ctx* ctx, const double* x, int64_t count, int64_t* first_bad) {
    if (ctx == nullptr) return VW_E_NULL;
    if (first_bad == nullptr) return set_err(ctx, VW_E_NULL, "first_bad cannot be null");
    *first_bad = -1;
    if (count < 0) return set_err(ctx, VW_E_NOT_POSITIVE, "negative count");
    if (count == 0) return VW_OK;
    if (x == nullptr) return set_err(ctx, VW_E_NULL, "signal cannot be null");
    VW_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // staging buffers are shared per context
    {
        int rc = ensure(ctx, ctx->stage_in, static_cast<size_t>(count) * 8);
        if (rc == VW_OK) rc = ensure(ctx, ctx->cell, 8);
        if (rc != VW_OK) return rc;
    }
    double* d = static_cast<double*>(ctx->stage_in.p);
    int64_t* cell = static_cast<int64_t*>(ctx->cell.p);
    VW_HIP(ctx, hipMemcpyAsync(d, x, static_cast<size_t>(count) * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = vw_first_non_finite_dev(ctx, d, count, cell, ctx->stream);
    if (rc != VW_OK) return rc;
    VW_HIP(ctx, hipMemcpyAsync(first_bad, cell, 8, hipMemcpyDeviceToHost, ctx->stream));
    VW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*first_bad >= 0) return set_err(ctx, VW_E_NONFINITE, "signal contains a non-finite value at index %lld", static_cast<long long>(*first_bad));
    return VW_OK;
}
