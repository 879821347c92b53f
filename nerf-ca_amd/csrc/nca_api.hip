// nca_api.hip -- the C ABI of include/nerfca_hip.h: argument checking, workspace planning and
// kernel sequencing.  No torch types, no allocation of device memory, no synchronisation
// (except nca_timing_read, which waits for the events it reports).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <string>
#include <vector>
#include "nca_kernels.hpp"
#include "nca_wide.hpp"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(NCA_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------- timing
namespace {
struct TimedSpan { hipEvent_t a, b; int kind; };
std::mutex g_tmu;
bool g_timing = false;
std::vector<TimedSpan> g_spans;
std::vector<hipEvent_t> g_free;
double g_ms[NCA_K_COUNT];
int64_t g_cnt[NCA_K_COUNT];

hipEvent_t get_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct Span {
    int kind; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on;
    Span(int k, hipStream_t s) : kind(k), st(s) {
        std::lock_guard<std::mutex> lk(g_tmu);
        on = g_timing;
        if (on) {       // events recorded into a stream capture become graph nodes and cannot be timed: skip them
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) on = false;
        }
        if (on) { a = get_event(); b = get_event(); on = a && b && hipEventRecord(a, st) == hipSuccess; }
    }
    ~Span() {
        if (!on) return;
        const bool ok = hipEventRecord(b, st) == hipSuccess;
        std::lock_guard<std::mutex> lk(g_tmu);
        if (ok) g_spans.push_back({a, b, kind});
        else { g_free.push_back(a); g_free.push_back(b); }
    }
};

void drain_spans() {
    for (auto& s : g_spans) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            g_ms[s.kind] += ms;
            g_cnt[s.kind] += 1;
        }
        g_free.push_back(s.a);
        g_free.push_back(s.b);
    }
    g_spans.clear();
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
#ifdef NCA_CU_OVERRIDE
        // (the CU-scaling experiment's build only -- tools/cu_scaling.sh builds it with tools/variant_build_all.sh cus "-DNCA_CU_OVERRIDE=1";
        // the shipped library reads no such variable: NCA_CUS=<k> sizes every persistent grid for k compute units -- how the kernels
        // scale with the part of the chip they occupy tells a per-CU bound from a chip-wide one)
        const char* e = getenv("NCA_CUS");
        if (e && atoi(e) > 0 && atoi(e) < n) n = atoi(e);
#endif
    }
    return n;
}

// A second stream of the calling thread for the one place where two launches of a backward are independent AND bound by different things:
// the static net's weight gradient (HBM-bound, ready once that net's dgrad launch is done) and the dynamic net's dgrad launch
// (issue-bound) -- NCA_OPT_OVERLAP_CUS.  Fork and join are events on the caller's stream, so inside a stream capture the side stream
// joins the capture and the captured graph carries the two branches.  Created at the thread's first overlapped backward OUTSIDE a
// capture (creating a stream inside a global-mode capture would fail it); until it exists the same launches run in a row on the caller's stream --
// the results do not depend on it.
struct Fork { hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool tried = false; };
// one per (calling thread, device): a thread that drives several devices gets a side stream on each
Fork* fork_get(hipStream_t st) {
    static thread_local std::vector<Fork> forks;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    if ((size_t)dev >= forks.size()) forks.resize(dev + 1);
    Fork& f = forks[dev];
    if (!f.tried) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
        f.tried = true;
        hipStream_t s = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess) {
            f.side = s; f.ev_fork = a; f.ev_join = b;
        } else {
            if (s) (void)hipStreamDestroy(s);
            if (a) (void)hipEventDestroy(a);
            (void)hipGetLastError();
        }
    }
    return f.side ? &f : nullptr;
}
// A fork that is open when the backward leaves early (an error between fork and join) is joined here: the caller's stream waits for the
// side stream, so a stream capture never ends with an unjoined branch (which would fail EndCapture with an unrelated message).
struct ForkScope {
    Fork* f = nullptr; hipStream_t st = nullptr; bool open = false;
    ~ForkScope() {
        if (f && open) {
            (void)hipEventRecord(f->ev_join, f->side);
            (void)hipStreamWaitEvent(st, f->ev_join, 0);
        }
    }
};
}  // namespace

extern "C" int nca_timing_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timing = on != 0;
    return NCA_OK;
}
extern "C" int nca_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_tmu);
    drain_spans();
    memset(g_ms, 0, sizeof(g_ms));
    memset(g_cnt, 0, sizeof(g_cnt));
    return NCA_OK;
}
extern "C" int nca_timing_read(int32_t kind, double* total_ms, int64_t* launches) {
    if (kind < 0 || kind >= NCA_K_COUNT) return fail(NCA_E_INVALID, "timing kind %d out of range", kind);
    std::lock_guard<std::mutex> lk(g_tmu);
    drain_spans();
    if (total_ms) *total_ms = g_ms[kind];
    if (launches) *launches = g_cnt[kind];
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- basics
extern "C" int nca_abi_version(void) { return NCA_ABI_VERSION; }

// ---------------------------------------------------------------------------------- tunables
namespace {
constexpr int64_t OPT_AUTO = INT64_MIN;
std::mutex g_omu;
int64_t g_opt[NCA_OPT_COUNT];
bool g_opt_init = false;
void opt_init_locked() {
    if (g_opt_init) return;
    for (int i = 0; i < NCA_OPT_COUNT; ++i) g_opt[i] = OPT_AUTO;
    const char* e = getenv("NCA_RESIDENT");
    if (e && e[0] == '0') g_opt[NCA_OPT_RESIDENT_MIN_TILES] = -1;
    else if (e && e[0] == 'f') g_opt[NCA_OPT_RESIDENT_MIN_TILES] = 0;
    e = getenv("NCA_STAGE_FP8");
    if (e && (e[0] == '0' || e[0] == '1')) g_opt[NCA_OPT_STAGE_FP8] = e[0] - '0';
    e = getenv("NCA_WGRAD_W");
    if (e && atoi(e) >= 100 && atoi(e) <= 200) g_opt[NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT] = atoi(e);
    e = getenv("NCA_OVERLAP_CUS");
    if (e && atoi(e) >= 0) g_opt[NCA_OPT_OVERLAP_CUS] = atoi(e);
    e = getenv("NCA_BF16_STORE");
    if (e && (e[0] == '0' || e[0] == '1')) g_opt[NCA_OPT_BF16_STORE] = e[0] - '0';
    g_opt_init = true;
}
// default of NCA_OPT_STAGE_FP8_MIN_TILES: see stage_fp8_for()
constexpr int64_t STAGE_FP8_DEFAULT_MIN_TILES = 0;
// default of NCA_OPT_OVERLAP_CUS (the sweep on MI355X)
constexpr int64_t OVERLAP_CUS_DEFAULT = 0;
// per-call options of the entry point this thread is in (NcaRays.plan_opts; CallOpts below sets and clears it)
thread_local const NcaPlanOpts* t_call_opts = nullptr;
int64_t opt_value(int opt) {
    if (t_call_opts) {
        const int64_t c = opt == NCA_OPT_STAGE_FP8 ? t_call_opts->stage_fp8 : opt == NCA_OPT_STAGE_FP8_MIN_TILES ? t_call_opts->stage_fp8_min_tiles
                        : opt == NCA_OPT_RESIDENT_MIN_TILES ? t_call_opts->resident_min_tiles : opt == NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT ? t_call_opts->wgrad_rebuild_weight_pct
                        : opt == NCA_OPT_OVERLAP_CUS ? t_call_opts->overlap_cus : opt == NCA_OPT_BF16_STORE ? t_call_opts->bf16_store : NCA_OPT_UNSET;
        if (c != NCA_OPT_UNSET) return c;
    }
    std::lock_guard<std::mutex> lk(g_omu);
    opt_init_locked();
    int64_t v = g_opt[opt];
    if (v == OPT_AUTO) {
        if (opt == NCA_OPT_STAGE_FP8) v = -1;                                            // by batch size (threshold 0: always)
        if (opt == NCA_OPT_STAGE_FP8_MIN_TILES) v = STAGE_FP8_DEFAULT_MIN_TILES;
        if (opt == NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT) v = 115;
        if (opt == NCA_OPT_OVERLAP_CUS) v = OVERLAP_CUS_DEFAULT;
        if (opt == NCA_OPT_BF16_STORE) v = 1;
        if (opt == NCA_OPT_RESIDENT_MIN_TILES) v = (int64_t)8 * NCA_WAVES * num_cus();    // (at 4 tiles per wave -- the reference's 1 024 x 500 batch -- resident and streaming tie)
    }
    return v;
}
struct CallOpts {          // RAII: the per-call options apply while an entry point that takes NcaRays runs on this thread
    explicit CallOpts(const NcaRays* r) { t_call_opts = r ? r->plan_opts : nullptr; }
    ~CallOpts() { t_call_opts = nullptr; }
};
int check_plan_opts(const NcaPlanOpts* o);
}  // namespace
// A forward store (8-bit staging) for a bf16 batch of this many wave tiles?  Otherwise the bf16 forward writes no store and the backward
// recomputes the layers -- bf16 operands everywhere, nothing in 8 bits.  (Decided by nca_render_store_bytes / the storing forward; the
// backward follows what the forward reported.)
static bool stage_fp8_for(int64_t wave_tiles) {
    const int64_t v = opt_value(NCA_OPT_STAGE_FP8);
    if (v >= 0) return v != 0;
    return wave_tiles >= opt_value(NCA_OPT_STAGE_FP8_MIN_TILES);
}
extern "C" int nca_get_option(int32_t opt, int64_t* value) {
    if (opt < 0 || opt >= NCA_OPT_COUNT || opt == NCA_OPT_RESERVED0) return fail(NCA_E_INVALID, "option %d out of range", opt);
    if (!value) return fail(NCA_E_INVALID, "value is NULL");
    *value = opt_value(opt);
    return NCA_OK;
}
static int check_option_value(int32_t opt, int64_t value) {
    if (opt < 0 || opt >= NCA_OPT_COUNT || opt == NCA_OPT_RESERVED0) return fail(NCA_E_INVALID, "option %d out of range", opt);
    if (opt == NCA_OPT_RESIDENT_MIN_TILES && value < -1) return fail(NCA_E_INVALID, "NCA_OPT_RESIDENT_MIN_TILES takes -1 (never), 0 (always) or a tile count");
    if (opt == NCA_OPT_STAGE_FP8 && value != 0 && value != 1 && value != -1) return fail(NCA_E_INVALID, "NCA_OPT_STAGE_FP8 takes 0 (nothing staged in 8 bits), 1 (always) or -1 (by batch size)");
    if (opt == NCA_OPT_STAGE_FP8_MIN_TILES && value < 0) return fail(NCA_E_INVALID, "NCA_OPT_STAGE_FP8_MIN_TILES takes a tile count >= 0");
    if (opt == NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT && (value < 100 || value > 200)) return fail(NCA_E_INVALID, "NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT takes 100 .. 200");
    if (opt == NCA_OPT_OVERLAP_CUS && (value < 0 || value > 4096)) return fail(NCA_E_INVALID, "NCA_OPT_OVERLAP_CUS takes 0 (off) or a number of compute units");
    if (opt == NCA_OPT_BF16_STORE && value != 0 && value != 1) return fail(NCA_E_INVALID, "NCA_OPT_BF16_STORE takes 0 (no store without 8-bit staging: recompute) or 1");
    return NCA_OK;
}
namespace {
int check_plan_opts(const NcaPlanOpts* o) {
    if (!o) return NCA_OK;
    const int64_t v[6] = {o->stage_fp8, o->stage_fp8_min_tiles, o->resident_min_tiles, o->wgrad_rebuild_weight_pct, o->overlap_cus, o->bf16_store};
    const int32_t k[6] = {NCA_OPT_STAGE_FP8, NCA_OPT_STAGE_FP8_MIN_TILES, NCA_OPT_RESIDENT_MIN_TILES, NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT, NCA_OPT_OVERLAP_CUS, NCA_OPT_BF16_STORE};
    for (int i = 0; i < 6; ++i)
        if (v[i] != NCA_OPT_UNSET) { int rc = check_option_value(k[i], v[i]); if (rc) return rc; }
    return NCA_OK;
}
}  // namespace
extern "C" int nca_set_option(int32_t opt, int64_t value) {
    int rc = check_option_value(opt, value);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_omu);
    opt_init_locked();
    g_opt[opt] = value;
    return NCA_OK;
}
extern "C" const char* nca_last_error(void) { return g_err; }

// Process-wide (a PyTorch backward runs on the autograd engine's thread, not on the thread that asks): the forward half is
// replaced by every nca_render_fwd, the backward half by every backward; each call fills a local copy and publishes it whole.
static NcaPlan g_plan_shared;
static std::mutex g_pmu;
static void merge_plan(NcaPlan& dst, const NcaPlan& pl, bool fwd) {
    if (fwd) {
        dst.fwd_store_format = pl.fwd_store_format; dst.fwd_launches = pl.fwd_launches; dst.fwd_resident = pl.fwd_resident;
    } else {
        const NcaPlan keep = dst;
        dst = pl;
        dst.fwd_store_format = keep.fwd_store_format; dst.fwd_launches = keep.fwd_launches; dst.fwd_resident = keep.fwd_resident;
    }
    dst.wave_tiles = pl.wave_tiles;
}
// `mine`: the caller's own record (NcaRays.plan_out) or null
static void publish_plan(const NcaPlan& pl, bool fwd, NcaPlan* mine = nullptr) {
    if (mine) merge_plan(*mine, pl, fwd);
    std::lock_guard<std::mutex> lk(g_pmu);
    merge_plan(g_plan_shared, pl, fwd);
}
extern "C" int nca_last_plan(NcaPlan* out) {
    if (!out) return fail(NCA_E_INVALID, "out is NULL");
    std::lock_guard<std::mutex> lk(g_pmu);
    *out = g_plan_shared;
    return NCA_OK;
}
extern "C" const char* nca_build_info(void) {
    // formatted once (thread-safe static initialisation); the compile-time experiment / variant / ablation masks are all 0 in a shipped
    // library (tests/test_host_cpu.py::test_shipped_library_is_not_a_timing_build).  The effective CU count is NOT part of the string
    // (reading it would initialise the device); the shipped library always sizes its grids for the whole chip (NCA_CU_OVERRIDE builds only
    // -- tools/cu_scaling.sh -- read NCA_CUS, and say so here).
    static const std::string info = [] {
        char b[160];
        snprintf(b, sizeof(b), "libnerfca_hip gfx950 abi=%d NCA_EXP=%d variant=0x%x ablation=0x%x%s", NCA_ABI_VERSION, nca_kernels_exp_mask(), nca_kernels_variant_mask(),
                 nca_kernels_ablation_mask(),
#ifdef NCA_CU_OVERRIDE
                 " cu-override"
#else
                 ""
#endif
        );
        return std::string(b);
    }();
    return info.c_str();
}

static int check_prec(int32_t prec) {
    if (prec != NCA_PREC_F32 && prec != NCA_PREC_BF16) return fail(NCA_E_UNSUPPORTED, "unknown precision %d", prec);
    return NCA_OK;
}

static int layout_of(const NcaNet* net, NcaLayout* y, int32_t prec = NCA_PREC_F32) {
    if (!net) return fail(NCA_E_INVALID, "net is NULL");
    const char* why = "";
    // f32 path: hidden-width contractions on the bf16 matrix cores from exact 3-way splits (NCA_F32_CHAIN=plain: the
    // v_mfma_f32_32x32x2_f32 loops, A/B switch)
    static const bool x3 = !(getenv("NCA_F32_CHAIN") != nullptr && getenv("NCA_F32_CHAIN")[0] == 'p');
    int rc = prec == NCA_PREC_BF16 ? nca_build_layout_bf16(*net, y, &why) : nca_build_layout(*net, y, &why, x3);
    if (rc != NCA_OK) return fail(rc, "%s", why);
    return NCA_OK;
}

// the general kernels (nets wider than 128 units / other channel counts: nca_wide.hpp); defined in nca_api_wide.inc behind run_bwd
struct NetBind;
static bool is_wide(const NcaNet* net);
static int wide_layout_of(const NcaNet* net, NcaWideLayout* y, int32_t prec);
static int64_t general_store_bytes(const NcaRays* rays, const NcaNet* const* nets, int nn, int32_t prec);

extern "C" int64_t nca_param_count(const NcaNet* net) {
    if (is_wide(net)) {
        NcaWideLayout w;
        int rc = wide_layout_of(net, &w, NCA_PREC_F32);
        return rc == NCA_OK ? w.n_params : rc;
    }
    NcaLayout y;
    int rc = layout_of(net, &y);
    return rc == NCA_OK ? y.n_params : rc;
}

extern "C" int64_t nca_packed_bytes(const NcaNet* net, int32_t prec) {
    NcaLayout y;
    int rc = check_prec(prec);
    if (rc) return rc;
    if (is_wide(net)) {
        NcaWideLayout w;
        rc = wide_layout_of(net, &w, prec);
        return rc == NCA_OK ? w.packed_floats * 4 : rc;
    }
    rc = layout_of(net, &y, prec);
    if (rc != NCA_OK) return rc;
    return y.packed_bytes;
}

extern "C" int nca_pack_weights(const NcaNet* net, const float* params, void* packed, int32_t prec, void* stream) {
    NcaLayout y;
    int rc = check_prec(prec);
    if (rc) return rc;
    if (is_wide(net)) {
        NcaWideLayout w;
        rc = wide_layout_of(net, &w, prec);
        if (rc) return rc;
        if (!params || !packed) return fail(NCA_E_INVALID, "params/packed is NULL");
        Span sp(NCA_K_PACK, (hipStream_t)stream);
        HIPCHK(nca_launch_wide_pack(w, params, static_cast<float*>(packed), (hipStream_t)stream));
        return NCA_OK;
    }
    rc = layout_of(net, &y, prec);
    if (rc != NCA_OK) return rc;
    if (!params || !packed) return fail(NCA_E_INVALID, "params/packed is NULL");
    Span sp(NCA_K_PACK, (hipStream_t)stream);
    if (prec == NCA_PREC_BF16) HIPCHK(nca_launch_pack_bf16(y, params, packed, (hipStream_t)stream));
    else HIPCHK(nca_launch_pack_f32(y, params, packed, (hipStream_t)stream));
    return NCA_OK;
}

extern "C" int nca_pack_weights2(const NcaNet* net_a, const float* params_a, void* packed_a,
                                 const NcaNet* net_b, const float* params_b, void* packed_b, int32_t prec, void* stream) {
    NcaLayout ya, yb;
    int rc = check_prec(prec);
    if (rc) return rc;
    if (is_wide(net_a) || is_wide(net_b)) {
        rc = nca_pack_weights(net_a, params_a, packed_a, prec, stream);
        return rc ? rc : nca_pack_weights(net_b, params_b, packed_b, prec, stream);
    }
    rc = layout_of(net_a, &ya, prec);
    if (rc != NCA_OK) return rc;
    rc = layout_of(net_b, &yb, prec);
    if (rc != NCA_OK) return rc;
    if (!params_a || !packed_a || !params_b || !packed_b) return fail(NCA_E_INVALID, "params/packed is NULL");
    Span sp(NCA_K_PACK, (hipStream_t)stream);
    if (prec == NCA_PREC_BF16) HIPCHK(nca_launch_pack2_bf16(ya, params_a, packed_a, yb, params_b, packed_b, (hipStream_t)stream));
    else HIPCHK(nca_launch_pack2_f32(ya, params_a, packed_a, yb, params_b, packed_b, (hipStream_t)stream));
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- helpers
static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct NetBind {
    const NcaNet* net; const void* packed; const float* win; const float* four; const float* params;
};

static int fill_net(const NetBind& b, NcaNetArgs* na, int32_t prec) {
    int rc = layout_of(b.net, &na->lay, prec);
    if (rc != NCA_OK) return rc;
    if (!b.packed) return fail(NCA_E_INVALID, "packed weights pointer is NULL");
    if (na->lay.enc_mode == NCA_ENC_BANDS && !b.win) return fail(NCA_E_INVALID, "band window is NULL");
    if (na->lay.enc_mode == NCA_ENC_FOURIER && !b.four) return fail(NCA_E_INVALID, "fourier coefficients are NULL");
    if (na->lay.T > 0 && !b.params) return fail(NCA_E_INVALID, "dynamic net needs its natural parameters (time latents)");
    na->win = b.win;
    na->four = b.four;
    na->lat = na->lay.T > 0 ? b.params + na->lay.lat_off : nullptr;
    na->row0 = 0;
    return NCA_OK;
}

static int add_stage(NcaFusedArgs* a, const void* base, uint32_t off, uint32_t bytes) {
    if (a->nstages >= NCA_MAX_STAGES) return fail(NCA_E_UNSUPPORTED, "too many weight stages");
    a->stage[a->nstages].ptr = static_cast<const char*>(base) + off;
    a->stage[a->nstages].bytes = (bytes + 1023u) & ~1023u;   // whole 1 KiB DMA pieces (the pack kernel zero-fills the padding)
    a->stage[a->nstages].lds_off = (uint32_t)a->res_total;       // resident layout: back to back
    a->res_total += (int32_t)((bytes + 15u) & ~15u);
    a->nstages++;
    return NCA_OK;
}

// stored: 0 = all forward images (+ dgrad images when bwd); 1 = backward from a store (it holds everything the backward needs:
// dgrad images only)
static int build_stages(NcaFusedArgs* a, const NetBind* binds, bool bwd, int stored = 0) {
    a->nstages = 0;
    a->res_total = 0;
    a->res_bytes = 0;
    for (int n = 0; n < a->nnets; ++n) {
        const NcaLayout& y = a->net[n].lay;
        for (int j = 0; stored != 1 && j < y.NL; ++j) {
            int rc = add_stage(a, binds[n].packed, y.layer[j].img_off, y.layer[j].img_bytes);
            if (rc) return rc;
            if (y.layer[j].img2_bytes) {            // second stage of a skip layer
                rc = add_stage(a, binds[n].packed, y.layer[j].img2_off, y.layer[j].img2_bytes);
                if (rc) return rc;
            }
        }
        if (bwd)
            for (int j = y.NL - 1; j >= 1; --j) {
                int rc = add_stage(a, binds[n].packed, y.layer[j].imgT_off, y.layer[j].imgT_bytes);
                if (rc) return rc;
                if (y.layer[j].imgT2_bytes) {          // x3: second sub-stage of the transposed image
                    rc = add_stage(a, binds[n].packed, y.layer[j].imgT2_off, y.layer[j].imgT2_bytes);
                    if (rc) return rc;
                }
            }
    }
    return NCA_OK;
}

// bf16: lay every weight image of the launch out in LDS, back to back.  False (and res_bytes = 0: the streaming kernel) when
// they do not fit next to the launch's other LDS needs or the batch is below NCA_OPT_RESIDENT_MIN_TILES.
static bool plan_resident(NcaFusedArgs* a, int kmode) {
    const uint32_t tight = (uint32_t)a->res_total;          // left by build_stages
    a->res_bytes = 0;
    const int64_t min_tiles = opt_value(NCA_OPT_RESIDENT_MIN_TILES);
    if (min_tiles < 0 || a->ntiles < min_tiles || a->nnets != 1) {
        if (getenv("NCA_DEBUG")) fprintf(stderr, "[nerfca] plan_resident: off (min tiles %lld, tiles %lld, nets %d)\n", (long long)min_tiles, (long long)a->ntiles, a->nnets);
        return false;
    }
    if (a->nstages <= 0) return false;
    if (nca_has_skip(a->net[0].lay)) return false;          // (a skip layer's two images take both halves of the streaming kernels' double buffer)
    const uint32_t dma_end = a->stage[a->nstages - 1].lds_off + a->stage[a->nstages - 1].bytes;   // the DMA moves whole 1 KiB pieces
    const size_t other = nca_fused_bf16_lds_other(a->net[0].lay.F, kmode);
    const size_t need = tight + other > dma_end ? tight + other : dma_end;
    static const bool dbg = getenv("NCA_DEBUG") != nullptr;
    const bool fits = need <= (size_t)NCA_LDS_BYTES;
    if (dbg) fprintf(stderr, "[nerfca] plan_resident: mode %d, %d stages, %u image bytes + %zu other: %s\n", kmode, a->nstages, tight, other,
                     fits ? "resident" : "streaming");
    if (!fits) return false;
    a->res_bytes = (int32_t)tight;
    return true;
}

static int check_rays(const NcaRays* r) {
    if (!r) return fail(NCA_E_INVALID, "rays is NULL");
    if (r->R <= 0 || r->S <= 0) return fail(NCA_E_INVALID, "empty ray batch (R=%lld, S=%d)", (long long)r->R, r->S);
    if (!r->origins || !r->dirs || !r->z || !r->dists || !r->I0) return fail(NCA_E_INVALID, "a ray input pointer is NULL");
    if (r->act < 0 || r->act > 2) return fail(NCA_E_INVALID, "unknown activation %d", r->act);
    return check_plan_opts(r->plan_opts);
}

static inline int tile_samples(int32_t prec) { return prec == NCA_PREC_BF16 ? 64 : 32; }

static void rays_to_args(const NcaRays* r, NcaFusedArgs* a, int32_t prec) {
    a->mode = NCA_MODE_RAYS;
    a->S = r->S;
    a->nchunk = (r->S + tile_samples(prec) - 1) / tile_samples(prec);
    a->ray_is_f64 = r->ray_is_f64;
    a->act = r->act;
    a->single = r->single_field;
    a->scale = r->scale;
    a->origins = r->origins;
    a->dirs = r->dirs;
    a->phase = r->phase;
    a->ps_r = r->phase_stride_r;
    a->ps_s = r->phase_stride_s;
    a->z = r->z;
    a->zs_r = r->z_stride_r;
    a->dists = r->dists;
}

// ---------------------------------------------------------------------------------- forward
extern "C" int nca_composite_fwd(int64_t, int32_t, int32_t, int32_t, float, const float*, const float*, const float*, const double*, double*, float*, float*, void*);
extern "C" int nca_composite_bwd(int64_t, int32_t, int32_t, int32_t, float, const float*, const float*, const double*, const double*, const float*, const float*, float*, float*, void*);
// The store a forward can leave behind for its backward (so that the backward does not recompute the layers):
//   H region  [32-sample tile][net][input block | layer outputs]         bf16 mode: the input block and the outputs of layers 0 .. NL-2 as
//                                                                          e4m3 (nca_layout.hpp); f32: the inputs of all NL layers + the last output
//   masks     [wave tile][2][mask layers][1 KiB | 512 B]                  ReLU bit masks: bf16 mode of every layer (NL: the backward recomputes
//                                                                          nothing), f32 of the hidden layers' inputs (NL - 1)
//   raw       [wave tile][2][64 | 32] f32                                 raw net outputs
// The tile count is rounded up to the 8 waves of a workgroup (bf16): slack slots for the waves of the last group.
// bf16, static + dynamic net of one width with the same encoding (mode, bands, the SAME window / coefficient vectors):
// the dynamic net's input block is a superset of the static one's and is stored once
static bool can_share_enc(const NcaFusedArgs& a, int32_t prec) {
    if (prec != NCA_PREC_BF16 || a.nnets != 2) return false;
    const NcaLayout& s = a.net[0].lay;
    const NcaLayout& d = a.net[1].lay;
    return s.T == 0 && s.F == d.F && s.enc_mode == d.enc_mode && s.L == d.L && s.Kenc == d.Kenc && a.net[0].win == a.net[1].win &&
           a.net[0].four == a.net[1].four;
}

struct StorePlan {
    int64_t h_stride;     // per 32-sample tile of the H region: bytes (bf16) / rows of 32 floats (f32)
    int64_t row0[2], off_m, off_r, bytes;
    int32_t mask_layers;
};
// wave_tiles: 64-sample tiles (bf16) / 32-sample tiles (f32)
// h8 (bf16 mode): the 8-bit staged store (layer inputs as e4m3) or the bf16 store (layer inputs as bf16 fragments: NCA_STORE_BF16); both
// hold the masks of every layer and the raw outputs -- the backward from either recomputes nothing
static bool store_plan(const NcaLayout* lays, int nnets, int32_t prec, int64_t wave_tiles, StorePlan* sp, bool share_enc = false, bool h8 = true) {
    if (nnets == 2 && lays[0].F != lays[1].F) return false;
    const bool bf = prec == NCA_PREC_BF16;
    if (!bf) h8 = false;
    const int64_t EB = nca_bf_ebytes(bf && h8);
    // bf16: slack tile slots up to the next multiple of the 8 waves of a workgroup -- a wave without a tile writes there, so
    // that the storing forward's hot loops need no store predicate
    if (bf) wave_tiles = (wave_tiles + NCA_WAVES - 1) / NCA_WAVES * NCA_WAVES;
    memset(sp, 0, sizeof(*sp));
    for (int n = 0; n < nnets; ++n) {
        if (lays[n].NL < 2) return false;                 // no hidden layer: nothing worth storing
        sp->row0[n] = sp->h_stride;
        if (bf) sp->h_stride += EB + nca_bf_hbytes(lays[n], h8);                      // inputs of layers 0 .. NL-1
        else sp->h_stride += lays[n].K0rows_pad + (int64_t)lays[n].NL * lays[n].F;
        const int ml = bf ? lays[n].NL : lays[n].NL - 1;        // bf16 mode: the masks of every layer (the backward recomputes none)
        if (ml > sp->mask_layers) sp->mask_layers = ml;
    }
    if (share_enc) {      // [dynamic: input block + hidden blocks][static: hidden blocks only]
        const int64_t dyn = EB + nca_bf_hbytes(lays[1], h8);
        sp->row0[1] = 0;
        sp->row0[0] = dyn - EB;       // so that row0 + EB is where the static net's first hidden block starts
        sp->h_stride -= EB;
    }
    if (bf) {
        sp->off_m = align_up(wave_tiles * 2 * sp->h_stride, 1024);
        sp->off_r = sp->off_m + wave_tiles * 2 * sp->mask_layers * 1024;       // raw outputs [wave tile][net][64] f32
        sp->bytes = align_up(sp->off_r + wave_tiles * 2 * 64 * 4, 256);
    } else {
        sp->off_m = align_up(wave_tiles * sp->h_stride * 32 * 4, 1024);
        sp->off_r = sp->off_m + wave_tiles * 2 * sp->mask_layers * 512;
        sp->bytes = align_up(sp->off_r + wave_tiles * 2 * 32 * 4, 256);
    }
    return true;
}

extern "C" int64_t nca_render_store_bytes(const NcaRays* rays, const NcaNet* net_s, const NcaNet* net_d, int32_t prec) {
    CallOpts co(rays);
    int rc = check_rays(rays);
    if (rc) return rc;
    rc = check_prec(prec);
    if (rc) return rc;
    NcaLayout lays[2];
    const int nn = rays->single_field ? 1 : 2;
    if (is_wide(net_s) || (nn == 2 && is_wide(net_d))) {          // the general kernels' store: every layer's output, f32 (0 when a fused-kernel net shares the batch)
        const NcaNet* nets[2] = {net_s, net_d};
        return general_store_bytes(rays, nets, nn, prec);
    }
    rc = layout_of(net_s, &lays[0], prec);
    if (rc) return rc;
    if (nn == 2) { rc = layout_of(net_d, &lays[1], prec); if (rc) return rc; }
    StorePlan sp;
    const int ts = tile_samples(prec);
    const int64_t wave_tiles = rays->R * ((rays->S + ts - 1) / ts);
    // bf16 mode: the 8-bit staged store, or -- NCA_OPT_STAGE_FP8 says "nothing in 8 bits" for this batch -- the bf16 store
    const bool h8 = prec == NCA_PREC_BF16 && stage_fp8_for(wave_tiles);
    if (prec == NCA_PREC_BF16 && !h8 && opt_value(NCA_OPT_BF16_STORE) == 0) return 0;          // no store at all: the backward will recompute
    if (!store_plan(lays, nn, prec, wave_tiles, &sp, false, h8)) return 0;
    return sp.bytes;          // (the layout with one input block per net: a shared one needs less)
}

extern "C" int64_t nca_render_fwd_workspace(const NcaRays* rays) {
    int rc = check_rays(rays);
    if (rc) return rc;
    return align_up(rays->R * ((rays->S + 31) / 32) * (int64_t)sizeof(double), 256);
}

static int64_t general_fwd_bytes(const NcaRays* rays, const NcaNet* const* nets, int nn, int32_t prec, int64_t max_bytes);
static int render_fwd_general(const NcaRays* rays, int32_t prec, const NetBind* binds, int nn, double* pix, float* sig_s, float* sig_d,
                              void* work, int64_t work_bytes, void* store, int64_t store_bytes, hipStream_t st);
extern "C" int64_t nca_render_fwd_workspace_nets(const NcaRays* rays, const NcaNet* net_s, const NcaNet* net_d, int32_t prec, int64_t max_bytes) {
    int rc = check_rays(rays);
    if (rc) return rc;
    rc = check_prec(prec);
    if (rc) return rc;
    const int64_t base = nca_render_fwd_workspace(rays);
    const NcaNet* nets[2] = {net_s, net_d};
    const int64_t gen = general_fwd_bytes(rays, nets, rays->single_field ? 1 : 2, prec, max_bytes);
    if (gen < 0) return gen;
    return gen > base ? gen : base;
}

extern "C" int nca_render_fwd(const NcaRays* rays, int32_t prec,
                              const NcaNet* net_s, const void* packed_s, const float* win_s, const float* four_s,
                              const NcaNet* net_d, const void* packed_d, const float* win_d, const float* four_d,
                              const float* latents_d,
                              double* pix, float* sig_s, float* sig_d, void* work, int64_t work_bytes,
                              void* store, int64_t store_bytes, void* stream) {
    CallOpts co(rays);
    int rc = check_rays(rays);
    if (rc) return rc;
    rc = check_prec(prec);
    if (rc) return rc;
    if (!sig_s || (!rays->single_field && !sig_d)) return fail(NCA_E_INVALID, "an output pointer is NULL");
    if (!rays->single_field && !net_d) return fail(NCA_E_INVALID, "composite render needs the dynamic net");
    const int64_t need = nca_render_fwd_workspace(rays);
    if (!work || work_bytes < need) return fail(NCA_E_WORKSPACE, "forward workspace %lld < %lld bytes", (long long)work_bytes, (long long)need);

    static thread_local NcaFusedArgs a;
    memset(&a, 0, sizeof(a));
    NcaPlan g_plan;
    memset(&g_plan, 0, sizeof(g_plan));
    rays_to_args(rays, &a, prec);
    a.nnets = rays->single_field ? 1 : 2;
    NetBind binds[2] = {{net_s, packed_s, win_s, four_s, nullptr}, {net_d, packed_d, win_d, four_d, latents_d}};
    if (is_wide(net_s) || (a.nnets == 2 && is_wide(net_d)))          // a net on the general kernels (more than 128 units): raw fields, then the compositing kernel
        return render_fwd_general(rays, prec, binds, a.nnets, pix, sig_s, sig_d, work, work_bytes, store, store_bytes, (hipStream_t)stream);
    for (int n = 0; n < a.nnets; ++n) {
        rc = fill_net(binds[n], &a.net[n], prec);
        if (rc) return rc;
    }
    if (a.net[0].lay.T > 0 && !rays->single_field) return fail(NCA_E_INVALID, "first net of a composite render must be static (T == 0)");
    for (int n = 0; n < a.nnets; ++n)
        if (a.net[n].lay.T > 0 && !rays->phase) return fail(NCA_E_INVALID, "dynamic net needs phase ids");
    a.ntiles = rays->R * a.nchunk;
    a.ray0 = 0;
    g_plan.wave_tiles = a.ntiles;
    const int64_t ngroups = (a.ntiles + NCA_WAVES - 1) / NCA_WAVES;
    const int grid = (int)(ngroups < num_cus() ? ngroups : num_cus());
    hipStream_t st = (hipStream_t)stream;
    if (a.nnets == 2 && a.net[0].lay.F != a.net[1].lay.F) {
        if (store) return fail(NCA_E_UNSUPPORTED, "a forward store needs nets of one width");
        if (!pix) return fail(NCA_E_UNSUPPORTED, "pix = NULL (ray sums left to the loss kernel) needs nets of one width");
        g_plan.fwd_launches = 2;
        publish_plan(g_plan, true, rays->plan_out);
        // nets of different width: one fused launch per net writes the raw field into its sigma buffer,
        // then the stand-alone compositing kernel turns both into sigmas + pix in place
        float* outs[2] = {sig_s, sig_d};
        for (int n = 0; n < 2; ++n) {
            static thread_local NcaFusedArgs one;
            one = a;
            one.nnets = 1;
            one.net[0] = a.net[n];
            one.raw_only = 1;
            one.raw_out = outs[n];
            NetBind b1[2] = {binds[n], {}};
            rc = build_stages(&one, b1, false);
            if (rc) return rc;
            if (prec == NCA_PREC_BF16) plan_resident(&one, NCA_KM_FWD);
            Span sp(NCA_K_FWD, st);
            if (prec == NCA_PREC_BF16) HIPCHK(nca_launch_fused_bf16(one.net[0].lay.F, one, false, grid, st));
            else HIPCHK(nca_launch_fused_f32(one.net[0].lay.F, one, false, grid, st));
        }
        return nca_composite_fwd(rays->R, rays->S, rays->act, 0, rays->scale, sig_s, sig_d, rays->I0, rays->dists, pix, sig_s, sig_d, stream);
    }
    rc = build_stages(&a, binds, false);
    if (rc) return rc;
    a.part = static_cast<double*>(work);
    a.sig_s = sig_s;
    a.sig_d = sig_d;
    int kmode = NCA_KM_FWD;
    int store_format = NCA_STORE_NONE;
    // bf16 with neither store allowed for this batch (NCA_OPT_STAGE_FP8 off, NCA_OPT_BF16_STORE = 0): the buffer is left untouched and the
    // return value says so (0) -- the backward recomputes
    if (store && prec == NCA_PREC_BF16 && !stage_fp8_for(a.ntiles) && opt_value(NCA_OPT_BF16_STORE) == 0) store = nullptr;
    if (store) {
        NcaLayout lays[2] = {a.net[0].lay, a.net[1].lay};
        StorePlan spl;
        a.share_enc = can_share_enc(a, prec) ? 1 : 0;
        // bf16 mode: layer inputs as e4m3 (the 8-bit staged store) or, with NCA_OPT_STAGE_FP8 saying "nothing in 8 bits" for this batch,
        // as bf16 fragments (NCA_STORE_BF16) -- the return value tells the backward which
        a.h8 = (prec == NCA_PREC_BF16 && stage_fp8_for(a.ntiles)) ? 1 : 0;
        store_format = (prec != NCA_PREC_BF16 ? NCA_STORE_F32 : a.h8 ? NCA_STORE_FP8 : NCA_STORE_BF16) | (a.share_enc ? NCA_STORE_SHARED_ENC : 0);
        if (!store_plan(lays, a.nnets, prec, a.ntiles, &spl, a.share_enc != 0, a.h8 != 0))
            return fail(NCA_E_UNSUPPORTED, "a forward store needs nets of one width with at least one hidden layer");
        if (store_bytes < spl.bytes) return fail(NCA_E_WORKSPACE, "forward store %lld < %lld bytes", (long long)store_bytes, (long long)spl.bytes);
        kmode = NCA_KM_FWD_STORE;
        a.scratch = static_cast<float*>(store);
        a.rows_total = spl.h_stride;
        for (int n = 0; n < a.nnets; ++n) a.net[n].row0 = spl.row0[n];
        a.tile0 = 0;
        a.mstore = static_cast<char*>(store) + spl.off_m;
        a.rstore = reinterpret_cast<float*>(static_cast<char*>(store) + spl.off_r);
        a.mstore_layers = spl.mask_layers;
    }
    bool done = false;
    if (prec == NCA_PREC_BF16 && a.nnets == 2) {
        // one launch per net with that net's weight images resident in LDS: the static net writes its sigma, the dynamic net
        // reads it back and composites
        static thread_local NcaFusedArgs one[2];
        bool ok = true;
        for (int n = 0; n < 2 && ok; ++n) {
            one[n] = a;
            one[n].nnets = 1;
            one[n].net[0] = a.net[n];
            one[n].net_base = n;
            one[n].split = n + 1;
            NetBind b1[2] = {binds[n], {}};
            rc = build_stages(&one[n], b1, false);
            if (rc) return rc;
            ok = plan_resident(&one[n], kmode);
        }
        if (ok) {
            for (int n = 0; n < 2; ++n) {
                Span sp(NCA_K_FWD, st);
                HIPCHK(nca_launch_fused_bf16(one[n].net[0].lay.F, one[n], kmode, grid, st, a.h8 != 0));
            }
            done = true;
            g_plan.fwd_launches = 2;
            g_plan.fwd_resident = 1;
        }
    } else if (prec == NCA_PREC_BF16) {
        g_plan.fwd_resident = plan_resident(&a, kmode) ? 1 : 0;
    }
    if (!done) {
        Span sp(NCA_K_FWD, st);
        if (prec == NCA_PREC_BF16) HIPCHK(nca_launch_fused_bf16(a.net[0].lay.F, a, kmode, grid, st, a.h8 != 0));
        else HIPCHK(nca_launch_fused_f32(a.net[0].lay.F, a, kmode, grid, st));
        g_plan.fwd_launches = 1;
    }
    if (pix) HIPCHK(nca_launch_pix_f32(rays->R, a.nchunk, rays->I0, a.part, pix, st));       // (NULL: the loss kernel forms pix from a.part, NcaLoss.ray_part)
    g_plan.fwd_store_format = store_format;
    publish_plan(g_plan, true, rays->plan_out);
    return store_format;
}

// ---------------------------------------------------------------------------------- backward
// A "unit" is what the batch is chunked by: one ray (rays mode) or one wave tile of points.
struct BwdPlan {
    int64_t tile_stride;   // f32: scratch rows per 32-sample tile; bf16: BYTES per 32-sample tile (all nets)
    int64_t slab_stride;   // floats per split slab
    int n_split, njobs, grid;
    int n_split_x;          // splits of the jobs that rebuild their D block from mask bits (more waves, fewer tiles each); slab rows = the larger
    int64_t units_per_chunk;
    int64_t tiles_per_unit;    // wave tiles (32 samples f32 / 64 samples bf16) per unit
    int64_t bytes_total;
    int64_t off_slab, off_oslab, off_scratch;
    // NCA_OPT_OVERLAP_CUS (two nets, bf16 mode 5, resident, e5m2 staging): one weight-gradient launch PER NET -- net 0's runs beside net 1's
    // dgrad launch on ovl_cus compute units, net 1's after the join on the whole chip; 0 = one launch for both nets after both dgrad launches
    int ovl_cus;
    int ns_net[2], nx_net[2];   // splits of the regular / the rebuilding jobs of each net's launch
    int grid_net[2];            // workgroups of each net's dgrad launch
};

static int64_t scratch_rows(const NcaLayout& y) { return y.K0rows_pad + (int64_t)(y.NL - 1) * y.F + (int64_t)y.NL * y.F; }

// d8: bf16 backward from a store with fp8 staging (D_0..D_{NL-2} as e5m2 + one inverse-scale record per tile)
static int plan_bwd(const NcaLayout* lays, int nnets, int32_t prec, int64_t units, int64_t tiles_per_unit, int64_t budget, BwdPlan* p,
                    bool stored = false, bool d8 = false, bool nr = false, int ovl_cus = 0) {
    const bool bf = prec == NCA_PREC_BF16;
    p->ovl_cus = 0;
    p->tile_stride = 0;
    p->slab_stride = 0;
    p->njobs = 0;
    for (int n = 0; n < nnets; ++n) {
        if (stored) p->tile_stride += bf ? nca_bf_dbytes(lays[n], d8) : (int64_t)lays[n].NL * lays[n].F;   // only the D blocks live in the chunk scratch
        else p->tile_stride += bf ? nca_bf_tile_bytes(lays[n]) : scratch_rows(lays[n]);
        p->slab_stride += lays[n].n_params;
        for (int j = 0; j < lays[n].NL; ++j) p->njobs += lays[n].layer[j].kind == NCA_IN_SKIP ? 2 : 1;
    }
    if (stored && bf && (d8 || nr)) p->tile_stride += NCA_D8_REC_BYTES;
    for (int n = 0; n < nnets; ++n) p->slab_stride += (int64_t)lays[n].F * lays[n].P;
    p->slab_stride = align_up(p->slab_stride, 64);
    if (p->njobs > NCA_MAX_JOBS) return fail(NCA_E_UNSUPPORTED, "too many wgrad jobs");
    const int cus = num_cus();
    p->tiles_per_unit = tiles_per_unit;
    const int F = lays[0].F;
    // f32 wgrad: 256-thread workgroups, two per CU.  bf16 wgrad: one wave per (job, split), four per CU.
    // all (split, job) blocks must be co-resident in ONE round: floor, never ceil (6 stragglers of 1030
    // blocks on 1024 slots double the kernel time)
    int nsplit = ((bf ? 4 : 2) * cus) / (p->njobs > 0 ? p->njobs : 1);
    if (nsplit < 1) nsplit = 1;
    // The jobs of the last hidden layers under e5m2 staging rebuild their D block on the vector ALU: 1.15 x the cycles per tile of
    // the others (measured with round 3's clock-probe build, tools/r03_experiments.sh).  The grid is ONE round of one-wave workgroups, so the slowest wave is the launch:
    // those jobs get W x the splits (NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT, default 1.15).  Their extra slab rows hold NOTHING in every other job's columns and are never read there (NcaReduceArgs::n_split_std).
    int nsplit_x = nsplit;
    int slab_rows = 0;          // rows of the split slab: the most splits any job of any launch of this plan runs over
    int nexp = 0;          // rebuilding jobs: one per net, two where the last layer is a skip layer (its encoded and its hidden part)
    for (int n = 0; n < nnets; ++n) nexp += lays[n].layer[lays[n].NL - 1].kind == NCA_IN_SKIP ? 2 : 1;
    if (bf && stored && d8 && nr && p->njobs > nexp) {
        const double W = 0.01 * (double)opt_value(NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT);
        int ns = (int)((4.0 * cus) / ((p->njobs - nexp) + nexp * W));
        if (ns < 1) ns = 1;
        int nx = (int)(W * ns);
        while ((p->njobs - nexp) * ns + nexp * nx > 4 * cus && nx > ns) --nx;
        nsplit = ns;
        nsplit_x = nx;
        // one launch per net: net 0's on ovl_cus compute units (beside net 1's dgrad launch on the others), net 1's on all of them
        if (ovl_cus >= 8 && ovl_cus <= cus - 8 && nnets == 2 && lays[0].NL > 1 && lays[1].NL > 1 && nexp == 2) {
            p->ovl_cus = ovl_cus;
            for (int n = 0; n < 2; ++n) {
                const int slots = 4 * (n == 0 ? ovl_cus : cus), nj = lays[n].NL;
                int s0 = (int)(slots / ((nj - 1) + W));
                if (s0 < 1) s0 = 1;
                int x0 = (int)(W * s0);
                while ((nj - 1) * s0 + x0 > slots && x0 > s0) --x0;
                p->ns_net[n] = s0;
                p->nx_net[n] = x0;
                if (x0 > slab_rows) slab_rows = x0;
            }
        }
    }
    if (nsplit_x > slab_rows) slab_rows = nsplit_x;
    const int64_t slab_bytes = align_up((int64_t)slab_rows * p->slab_stride * 4, 256);
    const int64_t oslab_bytes = align_up((int64_t)cus * 2 * (F + 1) * 4, 256);
    // bf16 backward from a store: slack tile slots behind the D region, for the waves of the last group that have no tile
    const int64_t dslack = (bf && stored) ? (int64_t)(NCA_WAVES - 1) * 2 * p->tile_stride : 0;
    const int64_t fixed = slab_bytes + oslab_bytes + align_up(dslack, 256);
    // scratch bytes per unit: f32 rows*32 floats per 32-sample tile; bf16 two 32-sample tiles per wave tile
    const int64_t per_unit = bf ? p->tile_stride * 2 * tiles_per_unit : p->tile_stride * tiles_per_unit * 32 * 4;
    int64_t upc = units;
    if (budget > 0) {
        int64_t avail = budget - fixed;
        int64_t fit = avail > 0 ? avail / per_unit : 0;
        if (fit < 1) fit = 1;
        if (fit < upc) upc = fit;
    }
    p->units_per_chunk = upc;
    const int64_t tiles = upc * tiles_per_unit;
    const int64_t ktiles = bf ? tiles * 2 : tiles;       // 32-sample tiles the wgrad splits over
    if ((int64_t)nsplit_x > ktiles / 2) { nsplit_x = nsplit < nsplit_x ? nsplit : nsplit_x; }      // small launches: uniform splits
    if ((int64_t)nsplit > ktiles) nsplit = (int)ktiles;
    if ((int64_t)nsplit_x > ktiles) nsplit_x = (int)ktiles;
    if (nsplit_x < nsplit) nsplit_x = nsplit;
    p->n_split = nsplit;
    p->n_split_x = nsplit_x;
    const int64_t ngroups = (tiles + NCA_WAVES - 1) / NCA_WAVES;
    p->grid = (int)(ngroups < cus ? ngroups : cus);
    if (p->ovl_cus) {
        // (the overlapped plan is for batches that fill the chip many times over: a launch whose clamps bite runs the plain plan)
        if (ktiles < 8 * (int64_t)slab_rows || ngroups < cus) p->ovl_cus = 0;
        else { p->grid_net[0] = cus; p->grid_net[1] = cus - p->ovl_cus; }
    }
    p->off_slab = 0;
    p->off_oslab = slab_bytes;
    p->off_scratch = slab_bytes + oslab_bytes;
    p->bytes_total = fixed + align_up(per_unit * upc, 256);
    return NCA_OK;
}

// row0: first row of the net's input block in a tile of the H region; drow0: of its D_0 in a tile of the D region
static void add_jobs_f32(NcaWgradArgs* w, const NcaLayout& y, int64_t row0, int64_t drow0, int64_t slab_off, int64_t onehot_off) {
    const int64_t hrow0 = row0 + y.K0rows_pad;                    // inputs of layers 1..NL-1
    for (int j = 0; j < y.NL; ++j) {
        const NcaLayerL& l = y.layer[j];
        bool bias_done = false;
        if (l.kind != NCA_IN_HID) {
            NcaWgradJob& g = w->job[w->njobs++];
            g.F = y.F;
            g.d_row0 = drow0 + (int64_t)j * y.F;
            g.b_row0 = row0;
            g.ncols_w = y.K0;
            g.P = (j == 0) ? y.P : 0;
            g.b_rows_pad = (int32_t)align_up(g.ncols_w + g.P, 32);
            g.out_off = slab_off + l.w_off;
            g.out_ld = l.K;
            g.out_col0 = 0;
            g.onehot_off = onehot_off;
            g.bias_off = slab_off + l.b_off;
            bias_done = true;
        }
        if (l.kind != NCA_IN_ENC) {
            NcaWgradJob& g = w->job[w->njobs++];
            g.F = y.F;
            g.d_row0 = drow0 + (int64_t)j * y.F;
            g.b_row0 = hrow0 + (int64_t)(j - 1) * y.F;
            g.b_frag = 1;
            g.ncols_w = y.F;
            g.P = 0;
            g.b_rows_pad = y.F;
            g.out_off = slab_off + l.w_off;
            g.out_ld = l.K;
            g.out_col0 = l.kind == NCA_IN_SKIP ? y.K0 : 0;
            g.onehot_off = 0;
            g.bias_off = bias_done ? -1 : slab_off + l.b_off;
        }
    }
}

// net_off: byte offset of the net's input/H blocks in a tile of the H region; d_off: of its D blocks in a tile of the D region
// h8 / d8: formats of the store's hidden blocks and of this launch's D blocks (fp8 staging)
// part: 0 = the layer's (only) job; for a skip layer 1 = its encoded part (D_j x input block -> columns 0 .. K0 - 1 of the weight, and the bias), 2 = its hidden
// part (D_j x H_{j-1} -> columns K0 ..): the two stored blocks cat[encoded input, h] is made of (model/CPPN.py:102-104)
static void make_job_bf16(NcaWgradJob& g, const NcaLayout& y, int j, int64_t net_off, int64_t d_off, int64_t slab_off, int64_t onehot_off, int64_t enc_off,
                          bool h8, bool d8, int part = 0) {
    const int64_t EB = nca_bf_ebytes(h8);
    const NcaLayerL& l = y.layer[j];
    memset(&g, 0, sizeof(g));
    g.F = y.F;
    g.d_row0 = d_off + nca_bf_doff(y, j, d8);
    g.d8 = d8 ? 1 : 0;
    g.out_off = slab_off + l.w_off;
    g.out_ld = l.K;
    g.out_col0 = 0;
    g.bias_off = slab_off + l.b_off;
    g.onehot_off = onehot_off;
    if (part == 2) g.bias_off = -1;          // (the encoded-part job of the layer carries the bias)
    if (j == 0 || part == 1) {
        g.is_enc = 1;
        g.b_row0 = enc_off;          // the input block (the other net's when it is shared)
        g.b_row_bytes = NCA_BF_ENCROWS * 2;
        g.h8 = h8 ? 1 : 0;           // fp8 staging: the input block is stored as e4m3 as well
        g.ncols_w = y.Kenc;
        g.T = y.T;
        g.P = part == 1 ? 0 : y.P;   // (the one-hot phase rows belong to layer 0's job; a net with a skip layer has no latents anyway)
        g.fourier_L = y.enc_mode == NCA_ENC_FOURIER ? y.L : 0;
    } else {
        g.is_enc = 0;
        g.b_row0 = net_off + EB + nca_bf_hoff(y, j - 1, h8);
        g.h8 = h8 ? 1 : 0;
        g.b_row_bytes = y.F * 2;
        g.ncols_w = y.F;
        g.T = 0;
        g.P = 0;
        if (part == 2) g.out_col0 = y.K0;
    }
}
static void add_jobs_bf16(NcaWgradArgs* w, int net_index, const NcaLayout& y, int64_t net_off, int64_t d_off, int64_t slab_off, int64_t onehot_off, int64_t enc_off,
                          bool h8, bool d8, int64_t dscale_off, int expand_layer = -1, int mask_layers = 0) {
    for (int j = 0; j < y.NL; ++j) {
        const bool skip = y.layer[j].kind == NCA_IN_SKIP;
        for (int part = skip ? 1 : 0; part <= (skip ? 2 : 0); ++part) {
            NcaWgradJob& g = w->job[w->njobs++];
            make_job_bf16(g, y, j, net_off, d_off, slab_off, onehot_off, enc_off, h8, d8, part);
            g.net = net_index;
            g.dscale_off = dscale_off;
            if (j == expand_layer) {      // mode 5, e5m2: the block is rebuilt from the forward's mask bits (nca_kernels.hpp)
                g.expand = 1;
                g.mask_off = ((int64_t)net_index * mask_layers + j) * 1024;
            }
        }
    }
}

static int run_bwd(NcaFusedArgs& a, int32_t prec, const NetBind* binds, int64_t units, int64_t tiles_per_unit, float* const* grads,
                   void* work, int64_t work_bytes, hipStream_t st, const void* store = nullptr, int64_t store_bytes = 0, float* g_depth = nullptr,
                   int32_t store_format = NCA_STORE_NONE, NcaPlan* plan_out = nullptr, float* g_latents = nullptr) {
    const bool bf = prec == NCA_PREC_BF16;
    NcaPlan g_plan;       // (published at the end; what the last forward decided stays on record)
    memset(&g_plan, 0, sizeof(g_plan));
    g_plan.wave_tiles = units * tiles_per_unit;
    NcaLayout lays[2];
    for (int n = 0; n < a.nnets; ++n) lays[n] = a.net[n].lay;
    if (g_depth) {
        if (a.mode != NCA_MODE_RAYS) return fail(NCA_E_INVALID, "depth gradients need a ray batch");
        for (int n = 0; n < a.nnets; ++n) {
            if (lays[n].Kenc > 96) return fail(NCA_E_UNSUPPORTED, "depth gradients: more than 96 encoded features");
        }
    }
    // a store left by the forward of the SAME batch: no recompute, the chunk scratch holds the D blocks only
    StorePlan spl;
    const bool stored = store != nullptr;
    // fp8 staging: the store's hidden blocks (as the storing forward left them) and this backward's output gradients -- except
    // when the depth gradient is wanted, whose kernel reads D_0 as bf16 fragments
    // The store's format is what the forward that wrote it reported (NcaRays.store_format), never this call's reading of the options
    if (stored) {
        const int kind = store_format & NCA_STORE_KIND_MASK;
        const bool kind_ok = bf ? (kind == NCA_STORE_FP8 || kind == NCA_STORE_BF16) : kind == NCA_STORE_F32;
        if ((store_format & ~(NCA_STORE_KIND_MASK | NCA_STORE_SHARED_ENC)) || !kind_ok)
            return fail(NCA_E_INVALID, "rays->store_format = %d does not name a store of this precision: pass the value nca_render_fwd returned when it wrote the store "
                                       "(0 = it wrote none: pass store = NULL)", store_format);
        const bool shared = (store_format & NCA_STORE_SHARED_ENC) != 0;
        if (shared != can_share_enc(a, prec))
            return fail(NCA_E_INVALID, "the store was written with %s input block, but the encoding vectors of this call say otherwise: pass the backward the SAME window / coefficient pointers as the forward",
                        shared ? "one shared" : "one per net");
    }
    // the bf16 mode's stores: layer inputs as e4m3 (8-bit staged: the output gradients then cross HBM as e5m2) or as bf16 fragments
    // (NCA_STORE_BF16: bf16 output gradients, the bf16 x bf16 weight-gradient jobs); both hold the masks of every layer and the raw outputs
    const bool h8 = bf && stored && (store_format & NCA_STORE_KIND_MASK) == NCA_STORE_FP8;
    const bool d8 = h8 && !g_depth;
    a.h8 = h8 ? 1 : 0;
    if (stored) {
        a.share_enc = (store_format & NCA_STORE_SHARED_ENC) ? 1 : 0;
        if (!store_plan(lays, a.nnets, prec, units * tiles_per_unit, &spl, a.share_enc != 0, h8)) return fail(NCA_E_UNSUPPORTED, "no forward store exists for this configuration");
        if (store_bytes < spl.bytes) return fail(NCA_E_WORKSPACE, "forward store %lld < %lld bytes", (long long)store_bytes, (long long)spl.bytes);
    }
    // the store holds every layer's output, masks and raw outputs -- mode 5, nothing recomputed
    const bool nr = bf && stored;
    // one launch per net with that net's weight images resident in LDS
    const int64_t res_min = opt_value(NCA_OPT_RESIDENT_MIN_TILES);
    bool res3 = bf && stored && res_min >= 0 && units * tiles_per_unit >= res_min;
    if (res3) {          // ... only if every net's images do fit (else: one launch for both nets, streaming)
        for (int n = 0; n < a.nnets && res3; ++n) {
            static thread_local NcaFusedArgs probe;
            probe = a;
            probe.nnets = 1;
            probe.net[0] = a.net[n];
            probe.ntiles = units * tiles_per_unit;
            NetBind b1[2] = {binds[n], {}};
            int rcp = build_stages(&probe, b1, true, 1);
            if (rcp) return rcp;
            res3 = plan_resident(&probe, NCA_KM_BWD_NR);
        }
    }
    const bool per_net_launch = res3;
    BwdPlan p;
    // (the overlapped plan: resident one-net launches of a ray batch with e5m2 staging -- the bench path; nothing else forks)
    const int ovl_opt = (per_net_launch && d8 && nr && a.mode == NCA_MODE_RAYS && !g_latents) ? (int)opt_value(NCA_OPT_OVERLAP_CUS) : 0;
    int rc = plan_bwd(lays, a.nnets, prec, units, tiles_per_unit, work_bytes, &p, stored, d8, nr, ovl_opt);
    if (rc) return rc;
    const bool ovl = p.ovl_cus > 0;
    Fork* fk = ovl ? fork_get(st) : nullptr;
    ForkScope fscope;
    fscope.f = fk; fscope.st = st;
    if (!work || work_bytes < p.bytes_total) return fail(NCA_E_WORKSPACE, "backward workspace %lld < %lld bytes", (long long)work_bytes, (long long)p.bytes_total);
    char* wb = static_cast<char*>(work);
    float* slab = reinterpret_cast<float*>(wb + p.off_slab);
    float* oslab = reinterpret_cast<float*>(wb + p.off_oslab);
    float* scratch = reinterpret_cast<float*>(wb + p.off_scratch);

    rc = build_stages(&a, binds, true, stored ? 1 : 0);
    if (rc) return rc;
    int64_t off = 0, soff = 0;
    for (int n = 0; n < a.nnets; ++n) { a.net[n].row0 = off; off += bf ? nca_bf_tile_bytes(lays[n]) : scratch_rows(lays[n]); }
    {     // where each net's D blocks start inside a tile of the D region; where the [Wo | bo] tail of its last image is
        int64_t doff = 0;
        for (int n = 0; n < a.nnets; ++n) {
            const NcaLayout& y = lays[n];
            const NcaLayerL& ll = y.layer[y.NL - 1];
            const char* pk = static_cast<const char*>(binds[n].packed);
            if (bf) {
                const int64_t EB = 32 * (int64_t)NCA_BF_ENCROWS * 2, HB = 32 * (int64_t)y.F * 2;
                if (stored) { a.net[n].row0 = spl.row0[n]; a.net[n].drow0 = doff; doff += nca_bf_dbytes(y, d8); }
                else a.net[n].drow0 = a.net[n].row0 + EB + (int64_t)(y.NL - 1) * HB;
                if (ll.kind == NCA_IN_SKIP) a.net[n].wo_src = reinterpret_cast<const float*>(pk + ll.img2_off + (int64_t)y.MT * (y.F / 16) * 1024);     // behind the k-steps of its second image
                else a.net[n].wo_src = reinterpret_cast<const float*>(pk + ll.img_off + (int64_t)y.MT * ll.ksteps * 1024) + 2 * y.MT * 16;
            } else {
                if (stored) { a.net[n].row0 = spl.row0[n]; a.net[n].drow0 = doff; doff += (int64_t)y.NL * y.F; }
                else a.net[n].drow0 = a.net[n].row0 + y.K0rows_pad + (int64_t)(y.NL - 1) * y.F;
                // f32 images: [k-steps x 64 x MT floats][bias 2 MT 16][Wo 2 MT 16][bo]; a last layer that is a skip
                // layer keeps [Wo | bo] behind the k-steps of its second (hidden-part) image
                if (ll.kind == NCA_IN_SKIP) a.net[n].wo_src = reinterpret_cast<const float*>(pk + ll.img2_off) + (int64_t)(ll.ksteps - ll.ksteps_enc) * 64 * y.MT;
                else if (y.x3 && ll.kind == NCA_IN_HID)   // x3 image: [Wo | bo] follows the fragments of the LAST sub-stage (after the bias if there is only one)
                    a.net[n].wo_src = ll.img2_bytes ? reinterpret_cast<const float*>(pk + ll.img2_off + nca_x3_sub_bytes(y.F))
                                                    : reinterpret_cast<const float*>(pk + ll.img_off + nca_x3_sub_bytes(y.F)) + 2 * y.MT * 16;
                else a.net[n].wo_src = reinterpret_cast<const float*>(pk + ll.img_off) + (int64_t)ll.ksteps * 64 * y.MT + 2 * y.MT * 16;
            }
        }
    }
    int64_t slab_off[2] = {0, 0}, onehot_off[2] = {0, 0};
    for (int n = 0; n < a.nnets; ++n) { slab_off[n] = soff; soff += lays[n].n_params; }
    for (int n = 0; n < a.nnets; ++n) { onehot_off[n] = soff; soff += (int64_t)lays[n].F * lays[n].P; }

    // the last hidden layer's output-gradient block is rebuilt by its weight-gradient job from the forward's mask bits (one byte / word per
    // sample instead of the block): with the 8-bit staged output gradients, and with bf16 ones out of the bf16 store -- not on the
    // depth-gradient path of the 8-bit store (bf16 D blocks beside e4m3 layer inputs: no such job) and not for points (per-point latents)
    const bool expand_last = nr && (d8 || (!h8 && !g_depth && !g_latents));
    a.expand_last = (expand_last && !d8) ? 1 : 0;
    static thread_local NcaWgradArgs w;
    memset(&w, 0, sizeof(w));
    for (int n = 0; n < a.nnets; ++n) {
        if (bf) add_jobs_bf16(&w, n, lays[n], a.net[n].row0, a.net[n].drow0, slab_off[n], onehot_off[n],
                              stored && a.share_enc && n == 0 ? a.net[1].row0 : a.net[n].row0, h8, d8,
                              p.tile_stride - NCA_D8_REC_BYTES, expand_last ? lays[n].NL - 1 : -1, spl.mask_layers);
        else add_jobs_f32(&w, lays[n], a.net[n].row0, a.net[n].drow0, slab_off[n], onehot_off[n]);
    }
    w.scratch = scratch;
    w.slab = slab;
    w.slab_stride = p.slab_stride;
    w.scratch_b = stored ? static_cast<const float*>(store) : scratch;
    w.rows_total_b = stored ? spl.h_stride : p.tile_stride;
    if (stored && bf) {
        w.mask = static_cast<const char*>(store) + spl.off_m;
        w.mask_stride = 2 * (int64_t)spl.mask_layers * 1024;
    }

    a.scratch = stored ? const_cast<float*>(static_cast<const float*>(store)) : scratch;
    a.dscratch = reinterpret_cast<char*>(scratch);
    a.d_total = p.tile_stride;
    a.dscale_off = p.tile_stride - NCA_D8_REC_BYTES;
    if (stored) {
        a.mstore = const_cast<char*>(static_cast<const char*>(store)) + spl.off_m;
        a.rstore = reinterpret_cast<float*>(const_cast<char*>(static_cast<const char*>(store)) + spl.off_r);
        a.mstore_layers = spl.mask_layers;
    }
    a.oslab = oslab;
    // bf16: keep the ReLU masks of the recomputed layers in LDS when they fit (8 KiB per layer per workgroup)
    a.mask_layers = 0;
    if (bf && !stored) {
        int ml = 0;
        for (int n = 0; n < a.nnets; ++n) if (lays[n].NL - 1 > ml) ml = lays[n].NL - 1;
        if (ml > 0 && ml <= 6) a.mask_layers = ml;
    }
    const int F = lays[0].F;
    const int wave_samples = tile_samples(prec);

    g_plan.bwd_kernel_mode = stored ? (nr ? NCA_KM_BWD_NR : NCA_KM_BWD_STORED) : NCA_KM_BWD;
    g_plan.bwd_onchip = 0;
    g_plan.stage_fp8 = h8 ? 1 : 0;
    g_plan.bwd_launches_per_chunk = (bf && stored && per_net_launch) ? a.nnets : 1;
    g_plan.wgrad_jobs = w.njobs;
    g_plan.wgrad_splits = ovl ? p.ns_net[1] : p.n_split;
    g_plan.wgrad_splits_rebuild = ovl ? p.nx_net[1] : p.n_split_x;
    g_plan.overlap_cus = p.ovl_cus;
    g_plan.overlap_forked = fk ? 1 : 0;
    // the weight-gradient launch of net n alone (overlapped plan): its jobs, its splits
    auto wgrad_net = [&](int n, hipStream_t ws, int waves_per_wg) -> hipError_t {
        static thread_local NcaWgradArgs w1;
        w1 = w;
        w1.njobs = 0;
        for (int j = 0; j < w.njobs; ++j)
            if (w.job[j].net == n) w1.job[w1.njobs++] = w.job[j];
        w1.nsplit_std = p.ns_net[n];
        w1.nsplit_x = p.nx_net[n];
        Span sp(NCA_K_BWD_WGRAD, ws);
        return nca_launch_wgrad_bf16(F, w1, p.nx_net[n], ws, waves_per_wg);
    };
    constexpr int ovl_nw = 4;          // (one 256-thread workgroup per CU: a CU holds EITHER kernel whichever is dispatched first)
    const bool one_chunk = units <= p.units_per_chunk;
    int chunk = 0;
    for (int64_t u0 = 0; u0 < units; u0 += p.units_per_chunk, ++chunk) {
        const int64_t nu = (u0 + p.units_per_chunk <= units) ? p.units_per_chunk : units - u0;
        a.ntiles = nu * tiles_per_unit;
        a.rows_total = stored ? spl.h_stride : p.tile_stride;
        a.tile0 = stored ? u0 * tiles_per_unit : 0;       // position of this chunk in the store of the whole batch
        a.accumulate = chunk > 0;
        if (a.mode == NCA_MODE_RAYS) a.ray0 = u0; else a.n0 = u0 * wave_samples;
        if (bf && stored && per_net_launch) {
            // one launch per net (nets are independent once the upstream gradients are known)
            for (int n = 0; n < a.nnets; ++n) {
                static thread_local NcaFusedArgs one;
                one = a;
                one.nnets = 1;
                one.net[0] = a.net[n];
                one.net_base = n;
                NetBind b1[2] = {binds[n], {}};
                const int km = NCA_KM_BWD_NR;
                rc = build_stages(&one, b1, true, 1);
                if (rc) return rc;
                if (res3) g_plan.bwd_resident = plan_resident(&one, km) ? 1 : 0;        // (does not fit: the streaming kernel, still one net per launch)
                {
                    Span sp(NCA_K_BWD_DGRAD, st);
                    HIPCHK(nca_launch_fused_bf16(F, one, km, ovl ? p.grid_net[n] : p.grid, st, d8));
                }
                if (ovl && n == 0) {
                    // FORK: net 0's output gradients are complete -- its weight gradient (HBM-bound, p.ovl_cus compute units' worth of
                    // one-round waves) runs beside net 1's dgrad launch (issue-bound, the other compute units)
                    w.rows_total = p.tile_stride;
                    w.tile0_b = u0 * tiles_per_unit * 2;
                    w.ntiles = a.ntiles * 2;
                    w.accumulate = chunk > 0;
                    hipStream_t ws = st;
                    if (fk) {
                        HIPCHK(hipEventRecord(fk->ev_fork, st));
                        HIPCHK(hipStreamWaitEvent(fk->side, fk->ev_fork, 0));
                        ws = fk->side;
                        fscope.open = true;
                    }
                    HIPCHK(wgrad_net(0, ws, ovl_nw));
                    if (fk) HIPCHK(hipEventRecord(fk->ev_join, fk->side));
                }
            }
        } else {
            Span sp(NCA_K_BWD_DGRAD, st);
            if (bf) HIPCHK(nca_launch_fused_bf16(F, a, stored ? (nr ? NCA_KM_BWD_NR : NCA_KM_BWD_STORED) : NCA_KM_BWD, p.grid, st, d8));
            else HIPCHK(nca_launch_fused_f32(F, a, stored ? NCA_KM_BWD_STORED : NCA_KM_BWD, p.grid, st));
        }
        if (nr && ovl) {        // (each net's partial rows are those of ITS dgrad launch's workgroups)
            for (int n = 0; n < a.nnets; ++n)
                HIPCHK(nca_launch_sum_tile_records(reinterpret_cast<const char*>(scratch), 2 * p.tile_stride, p.tile_stride - NCA_D8_REC_BYTES, a.ntiles, n, n + 1, F, oslab, p.grid_net[n], st));
        } else if (nr && !one_chunk) HIPCHK(nca_launch_sum_tile_records(reinterpret_cast<const char*>(scratch), 2 * p.tile_stride, p.tile_stride - NCA_D8_REC_BYTES, a.ntiles, 0, a.nnets, F, oslab, p.grid, st));
        // (one chunk: the records are summed by extra workgroups of the reduce launch below -- the D region is still this chunk's then)
        if (g_latents) {     // d loss / d latent input per point, from the same D_0 blocks (points mode, one net)
            NcaLatgradArgs lg;
            memset(&lg, 0, sizeof(lg));
            lg.ntiles = bf ? 2 * a.ntiles : a.ntiles;
            lg.n0 = u0 * wave_samples;
            lg.N = a.N;
            lg.T = lays[0].T; lg.Kenc = lays[0].Kenc; lg.ldw = lays[0].layer[0].K; lg.bf16 = bf ? 1 : 0;
            lg.w0 = binds[0].params + lays[0].layer[0].w_off;
            lg.dscratch = scratch; lg.d_total = p.tile_stride;
            lg.drow = a.net[0].drow0 + (bf ? nca_bf_doff(lays[0], 0, false) : 0);
            lg.g_lat = g_latents;
            HIPCHK(nca_launch_latgrad_f32(F, lg, st));
        }
        if (g_depth) {       // d loss / d depth from the D_0 blocks this chunk's dgrad launch just wrote
            NcaZgradArgs zg;
            memset(&zg, 0, sizeof(zg));
            zg.nnets = a.nnets; zg.S = a.S; zg.ray_is_f64 = a.ray_is_f64;
            zg.bf16 = bf ? 1 : 0;
            zg.nchunk = bf ? 2 * a.nchunk : a.nchunk;          // 32-sample tiles per ray (a bf16 wave tile is two of them)
            zg.ntiles = bf ? 2 * a.ntiles : a.ntiles;
            zg.ray0 = u0;
            zg.origins = a.origins; zg.dirs = a.dirs; zg.z = a.z; zg.zs_r = a.zs_r;
            zg.dscratch = scratch; zg.d_total = p.tile_stride;
            zg.g_z = g_depth;
            for (int n = 0; n < a.nnets; ++n) {
                NcaZgradNet& zn = zg.net[n];
                zn.F = lays[n].F; zn.enc_mode = lays[n].enc_mode; zn.L = lays[n].L; zn.Kenc = lays[n].Kenc;
                zn.win = a.net[n].win; zn.four = a.net[n].four;
                for (int j = 0; j < lays[n].NL; ++j) {       // every layer that reads the encoded input: layer 0 and a skip layer
                    if (lays[n].layer[j].kind == NCA_IN_HID || zn.nsrc >= 2) continue;
                    zn.w[zn.nsrc] = binds[n].params + lays[n].layer[j].w_off;
                    zn.ldw[zn.nsrc] = lays[n].layer[j].K;
                    zn.drow[zn.nsrc] = a.net[n].drow0 + (bf ? nca_bf_doff(lays[n], j, false) : (int64_t)j * lays[n].F);   // bf16: bytes (never e5m2 here: d8 is off)
                    ++zn.nsrc;
                }
            }
            HIPCHK(nca_launch_zgrad_f32(zg, st));
        }
        w.rows_total = p.tile_stride;
        w.tile0_b = stored ? u0 * tiles_per_unit * (bf ? 2 : 1) : 0;      // in 32-sample tiles
        w.ntiles = bf ? a.ntiles * 2 : a.ntiles;
        w.accumulate = chunk > 0;
        if (ovl) {
            // JOIN, then net 1's weight gradient on the whole chip
            if (fk) { HIPCHK(hipStreamWaitEvent(st, fk->ev_join, 0)); fscope.open = false; }
            HIPCHK(wgrad_net(1, st, 4));
        } else {
            Span sp(NCA_K_BWD_WGRAD, st);
            if (bf) {
                if (w.njobs) {
                    w.nsplit_std = p.n_split;
                    w.nsplit_x = p.n_split_x;
                    // (slab rows n_split .. n_split_x - 1 are written by the rebuilding jobs only, in their own columns; the reduce
                    // kernels sum every column over the rows its job wrote -- NcaReduceArgs::n_split_std -- so nothing has to be
                    // cleared.  A hipMemsetAsync of those rows used to stand here: captured into a HIP graph it left them unwritten
                    // and the replayed step added whatever the memory held to the gradient, tools/determinism_probe.py)
                    // the same one-wave (job, split) units as 256-thread workgroups, one per CU instead of four one-wave
                    // workgroups -- placement is then the same every launch (one-wave workgroups go round the XCDs and land unevenly from
                    // launch to launch): weight gradient 4.77 +- 0.09 -> 4.68 +- 0.01 ms, step 13.23 -> 13.13 ms over eight alternating runs
                    // on one box (profiles/r05_ab_wgrad_workgroups.txt); the bits do not change
                    HIPCHK(nca_launch_wgrad_bf16(F, w, p.n_split_x, st, 4));
                }
            } else HIPCHK(nca_launch_wgrad_f32(w, p.n_split, st));
        }
    }
    NcaReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.slab = slab;
    r.slab_stride = p.slab_stride;
    r.oslab = oslab;
    r.oslab_stride = 2 * (F + 1);
    for (int n = 0; n < a.nnets; ++n) {
        r.n_params[n] = lays[n].n_params;
        r.n_total += lays[n].n_params;
        NcaReduceNet& rn = r.net[n];
        rn.grads = grads[n];
        rn.params = binds[n].params;
        rn.slab_off = slab_off[n];
        rn.onehot_off = onehot_off[n];
        rn.F = lays[n].F; rn.T = lays[n].T; rn.P = lays[n].P; rn.K0 = lays[n].K0; rn.Kenc = lays[n].Kenc;
        rn.w0_off = lays[n].layer[0].w_off;
        rn.lat_count = (int64_t)lays[n].P * lays[n].T;
        rn.wo_off = lays[n].wo_off;
        rn.tail_from_sums = nr ? 1 : 0;
        rn.tl_w_off = lays[n].layer[lays[n].NL - 1].w_off;
        rn.tl_b_off = lays[n].layer[lays[n].NL - 1].b_off;
        rn.tl_K = lays[n].layer[lays[n].NL - 1].K;          // (F, or K0 + F where that layer is a skip layer)
        rn.n_split = ovl ? p.nx_net[n] : (bf ? p.n_split_x : p.n_split);
        rn.n_split_std = ovl ? p.ns_net[n] : p.n_split;
        rn.n_wg = ovl ? p.grid_net[n] : p.grid;
    }
    if (nr && !ovl && one_chunk) {
        r.rec_region = reinterpret_cast<const char*>(scratch);
        r.rec_tile_bytes = 2 * p.tile_stride; r.rec_off = p.tile_stride - NCA_D8_REC_BYTES; r.rec_ntiles = units * tiles_per_unit;
        r.rec_net0 = 0; r.rec_net1 = a.nnets; r.rec_F = F; r.rec_nwg = p.grid; r.rec_oslab = oslab;
    }
    {
        Span sp(NCA_K_BWD_REDUCE, st);
        HIPCHK(nca_launch_reduce_f32(r, st));
    }
    g_plan.chunks = chunk;
    publish_plan(g_plan, false, plan_out);
    return NCA_OK;
}

#include "nca_api_wide.inc"

extern "C" int64_t nca_render_bwd_workspace(const NcaRays* rays, const NcaNet* net_s, const NcaNet* net_d, int32_t prec, int64_t max_bytes) {
    CallOpts co(rays);
    int rc = check_rays(rays);
    if (rc) return rc;
    rc = check_prec(prec);
    if (rc) return rc;
    NcaLayout lays[2];
    int nn = rays->single_field ? 1 : 2;
    if (is_wide(net_s) || (nn == 2 && is_wide(net_d))) {
        const NcaNet* nets[2] = {net_s, net_d};
        return general_bwd_bytes(rays, nets, nn, prec, max_bytes);
    }
    rc = layout_of(net_s, &lays[0], prec);
    if (rc) return rc;
    if (nn == 2) { rc = layout_of(net_d, &lays[1], prec); if (rc) return rc; }
    BwdPlan p;
    const int ts = tile_samples(prec);
    if (nn == 2 && lays[0].F != lays[1].F) {
        // nets of different width run their backward one after the other in the same region; four [R,S] f32
        // arrays (raw_s, raw_d, g_raw_s, g_raw_d) sit behind it
        const int64_t extra = 4 * align_up(rays->R * (int64_t)rays->S * 4, 256);
        int64_t most = 0;
        for (int n = 0; n < 2; ++n) {
            rc = plan_bwd(&lays[n], 1, prec, rays->R, (rays->S + ts - 1) / ts, max_bytes > extra ? max_bytes - extra : 1, &p);
            if (rc) return rc;
            if (p.bytes_total > most) most = p.bytes_total;
        }
        return most + extra;
    }
    rc = plan_bwd(lays, nn, prec, rays->R, (rays->S + ts - 1) / ts, max_bytes, &p);
    if (rc) return rc;
    int64_t need = p.bytes_total;
    if (prec == NCA_PREC_BF16) {       // from a store: a record per tile, e5m2 or (depth gradients) bf16 output-gradient blocks
        // (run_bwd takes the overlapped plan only where it is eligible -- resident one-net launches of a ray batch -- and the plain one otherwise; under a
        // byte budget the two chunk differently, so the query covers both)
        const int ovl_opts[2] = {(int)opt_value(NCA_OPT_OVERLAP_CUS), 0};
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && ovl_opts[0] == 0) break;
            rc = plan_bwd(lays, nn, prec, rays->R, (rays->S + ts - 1) / ts, max_bytes, &p, true, true, true, ovl_opts[k]);
            if (rc) return rc;
            if (p.bytes_total > need) need = p.bytes_total;
        }
        rc = plan_bwd(lays, nn, prec, rays->R, (rays->S + ts - 1) / ts, max_bytes, &p, true, false, true);
        if (rc) return rc;
        if (p.bytes_total > need) need = p.bytes_total;
    }
    return need;
}

extern "C" int nca_render_bwd(const NcaRays* rays, int32_t prec,
                              const NcaNet* net_s, const void* packed_s, const float* win_s, const float* four_s, const float* params_s,
                              const NcaNet* net_d, const void* packed_d, const float* win_d, const float* four_d, const float* params_d,
                              const double* g_pix, const float* g_sig_s, const float* g_sig_d,
                              float* grads_s, float* grads_d, void* work, int64_t work_bytes,
                              const void* store, int64_t store_bytes, void* stream) {
    return nca_render_bwd_depth(rays, prec, net_s, packed_s, win_s, four_s, params_s, net_d, packed_d, win_d, four_d, params_d,
                                g_pix, g_sig_s, g_sig_d, grads_s, grads_d, nullptr, work, work_bytes, store, store_bytes, stream);
}

extern "C" int nca_render_bwd_depth(const NcaRays* rays, int32_t prec,
                              const NcaNet* net_s, const void* packed_s, const float* win_s, const float* four_s, const float* params_s,
                              const NcaNet* net_d, const void* packed_d, const float* win_d, const float* four_d, const float* params_d,
                              const double* g_pix, const float* g_sig_s, const float* g_sig_d,
                              float* grads_s, float* grads_d, float* g_depth, void* work, int64_t work_bytes,
                              const void* store, int64_t store_bytes, void* stream) {
    CallOpts co(rays);
    int rc = check_rays(rays);
    if (rc) return rc;
    rc = check_prec(prec);
    if (rc) return rc;
    if (!g_pix) return fail(NCA_E_INVALID, "g_pix is NULL");
    if (!grads_s || (!rays->single_field && !grads_d)) return fail(NCA_E_INVALID, "a gradient output pointer is NULL");
    if (!params_s || (!rays->single_field && !params_d)) return fail(NCA_E_INVALID, "natural parameters are required");
    static thread_local NcaFusedArgs a;
    memset(&a, 0, sizeof(a));
    rays_to_args(rays, &a, prec);
    a.nnets = rays->single_field ? 1 : 2;
    NetBind binds[2] = {{net_s, packed_s, win_s, four_s, params_s}, {net_d, packed_d, win_d, four_d, params_d}};
    if (is_wide(net_s) || (a.nnets == 2 && is_wide(net_d))) {
        float* gr[2] = {grads_s, grads_d};
        return render_bwd_general(rays, prec, binds, a.nnets, g_pix, g_sig_s, g_sig_d, gr, g_depth, work, work_bytes, store, store_bytes, (hipStream_t)stream);
    }
    for (int n = 0; n < a.nnets; ++n) {
        rc = fill_net(binds[n], &a.net[n], prec);
        if (rc) return rc;
    }
    for (int n = 0; n < a.nnets; ++n)
        if (a.net[n].lay.T > 0 && !rays->phase) return fail(NCA_E_INVALID, "dynamic net needs phase ids");
    float* grads[2] = {grads_s, grads_d};
    hipStream_t st = (hipStream_t)stream;
    if (a.nnets == 2 && a.net[0].lay.F != a.net[1].lay.F) {
        if (store) return fail(NCA_E_UNSUPPORTED, "a forward store needs nets of one width");
        if (g_depth) return fail(NCA_E_UNSUPPORTED, "depth gradients need nets of one width");
        // different widths: recompute both raw fields, push (g_pix, g_sigma) through the compositing chain rule
        // once, then run each net's backward on its own with the per-sample raw gradient
        const int64_t arr = align_up(rays->R * (int64_t)rays->S * 4, 256);
        if (!work || work_bytes < 4 * arr) return fail(NCA_E_WORKSPACE, "backward workspace too small");
        const int64_t region = work_bytes - 4 * arr;
        char* tailp = static_cast<char*>(work) + region;
        float* raw[2] = {reinterpret_cast<float*>(tailp), reinterpret_cast<float*>(tailp + arr)};
        float* graw[2] = {reinterpret_cast<float*>(tailp + 2 * arr), reinterpret_cast<float*>(tailp + 3 * arr)};
        a.ntiles = rays->R * a.nchunk;
        const int64_t ngroups = (a.ntiles + NCA_WAVES - 1) / NCA_WAVES;
        const int grid = (int)(ngroups < num_cus() ? ngroups : num_cus());
        for (int n = 0; n < 2; ++n) {
            static thread_local NcaFusedArgs one;
            one = a;
            one.nnets = 1;
            one.net[0] = a.net[n];
            one.raw_only = 1;
            one.raw_out = raw[n];
            NetBind b1[2] = {binds[n], {}};
            rc = build_stages(&one, b1, false);
            if (rc) return rc;
            if (prec == NCA_PREC_BF16) plan_resident(&one, NCA_KM_FWD);
            Span sp(NCA_K_FWD, st);
            if (prec == NCA_PREC_BF16) HIPCHK(nca_launch_fused_bf16(one.net[0].lay.F, one, false, grid, st));
            else HIPCHK(nca_launch_fused_f32(one.net[0].lay.F, one, false, grid, st));
        }
        rc = nca_composite_bwd(rays->R, rays->S, rays->act, 0, rays->scale, raw[0], raw[1], rays->dists, g_pix, g_sig_s, g_sig_d,
                               graw[0], graw[1], stream);
        if (rc) return rc;
        for (int n = 0; n < 2; ++n) {
            static thread_local NcaFusedArgs one;
            one = a;
            one.nnets = 1;
            one.net[0] = a.net[n];
            one.g_raw = graw[n];
            NetBind b1[2] = {binds[n], {}};
            float* g1[2] = {grads[n], nullptr};
            rc = run_bwd(one, prec, b1, rays->R, a.nchunk, g1, work, region, st);
            if (rc) return rc;
        }
        return NCA_OK;
    }
    a.g_pix = g_pix;
    a.g_sig_s = g_sig_s;
    a.g_sig_d = g_sig_d;
    return run_bwd(a, prec, binds, rays->R, a.nchunk, grads, work, work_bytes, st, store, store_bytes, g_depth, rays->store_format, rays->plan_out);
}

// ---------------------------------------------------------------------------------- point path
extern "C" int nca_mlp_fwd(const NcaNet* net, int32_t prec, const void* packed, const float* win, const float* four,
                           const float* params, int64_t N, const float* pts, const int32_t* phase, float* raw, void* stream) {
    int rc = check_prec(prec);
    if (rc) return rc;
    if (N <= 0) return fail(NCA_E_INVALID, "empty point batch");
    if (!pts || !raw) return fail(NCA_E_INVALID, "pts/raw is NULL");
    if (is_wide(net)) return fail(NCA_E_WORKSPACE, "a net on the general kernels (more than 128 units, or other channels than 3 -> 1) needs a workspace: call nca_mlp_fwd_ws");
    static thread_local NcaFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = NCA_MODE_POINTS;
    a.nnets = 1;
    NetBind binds[2] = {{net, packed, win, four, params}, {}};
    rc = fill_net(binds[0], &a.net[0], prec);
    if (rc) return rc;
    if (a.net[0].lay.T > 0 && !phase) return fail(NCA_E_INVALID, "dynamic net needs phase ids");
    rc = build_stages(&a, binds, false);
    if (rc) return rc;
    const int ts = tile_samples(prec);
    a.N = N;
    a.n0 = 0;
    a.pts = pts;
    a.phase = phase;
    a.raw_out = raw;
    a.ntiles = (N + ts - 1) / ts;
    const int64_t ngroups = (a.ntiles + NCA_WAVES - 1) / NCA_WAVES;
    const int grid = (int)(ngroups < num_cus() ? ngroups : num_cus());
    if (prec == NCA_PREC_BF16) plan_resident(&a, NCA_KM_FWD);
    Span sp(NCA_K_FWD, (hipStream_t)stream);
    if (prec == NCA_PREC_BF16) HIPCHK(nca_launch_fused_bf16(a.net[0].lay.F, a, false, grid, (hipStream_t)stream));
    else HIPCHK(nca_launch_fused_f32(a.net[0].lay.F, a, false, grid, (hipStream_t)stream));
    return NCA_OK;
}

static void points_to_geom(const float* pts, const int32_t* phase, NcaWideGeom* g) {
    memset(g, 0, sizeof(*g));
    g->mode = NCA_MODE_POINTS;
    g->pts = pts;
    g->phase = phase;
}
extern "C" int64_t nca_mlp_fwd_workspace(const NcaNet* net, int32_t prec, int64_t N, int64_t max_bytes) {
    int rc = check_prec(prec);
    if (rc) return rc;
    if (N <= 0) return fail(NCA_E_INVALID, "empty point batch");
    if (!is_wide(net)) return 0;
    NcaWideLayout y;
    rc = wide_layout_of(net, &y, prec);
    if (rc) return rc;
    WidePlan p;
    wide_plan(y, N, false, max_bytes, &p);
    return p.bytes;
}
extern "C" int nca_mlp_fwd_ws(const NcaNet* net, int32_t prec, const void* packed, const float* win, const float* four,
                              const float* params, int64_t N, const float* pts, const int32_t* phase, float* raw,
                              void* work, int64_t work_bytes, void* stream) {
    if (!is_wide(net)) return nca_mlp_fwd(net, prec, packed, win, four, params, N, pts, phase, raw, stream);
    int rc = check_prec(prec);
    if (rc) return rc;
    if (N <= 0) return fail(NCA_E_INVALID, "empty point batch");
    if (!pts || !raw) return fail(NCA_E_INVALID, "pts/raw is NULL");
    WideBind w;
    NetBind b{net, packed, win, four, params};
    rc = wide_bind(b, prec, &w);
    if (rc) return rc;
    NcaWideGeom geom;
    points_to_geom(pts, phase, &geom);
    return wide_forward(w, geom, N, raw, work, work_bytes, (hipStream_t)stream);
}

extern "C" int64_t nca_mlp_bwd_workspace(const NcaNet* net, int32_t prec, int64_t N, int64_t max_bytes) {
    int rc = check_prec(prec);
    if (rc) return rc;
    if (N <= 0) return fail(NCA_E_INVALID, "empty point batch");
    if (is_wide(net)) {
        NcaWideLayout y;
        rc = wide_layout_of(net, &y, prec);
        if (rc) return rc;
        WidePlan p;
        wide_plan(y, N, true, max_bytes, &p);
        return p.bytes;
    }
    NcaLayout lay;
    rc = layout_of(net, &lay, prec);
    if (rc) return rc;
    BwdPlan p;
    const int ts = tile_samples(prec);
    rc = plan_bwd(&lay, 1, prec, (N + ts - 1) / ts, 1, max_bytes, &p);
    if (rc) return rc;
    return p.bytes_total;
}

extern "C" int nca_mlp_bwd(const NcaNet* net, int32_t prec, const void* packed, const float* win, const float* four,
                           const float* params, int64_t N, const float* pts, const int32_t* phase, const float* g_raw,
                           float* grads, float* g_latents, void* work, int64_t work_bytes, void* stream) {
    int rc = check_prec(prec);
    if (rc) return rc;
    if (N <= 0) return fail(NCA_E_INVALID, "empty point batch");
    if (!pts || !g_raw || !grads || !params) return fail(NCA_E_INVALID, "a pointer is NULL");
    if (is_wide(net)) {
        if (g_latents) return fail(NCA_E_UNSUPPORTED, "per-point latent gradients of a net on the general kernels (the table-row sums in `grads` are complete)");
        WideBind w;
        NetBind b{net, packed, win, four, params};
        rc = wide_bind(b, prec, &w);
        if (rc) return rc;
        NcaWideGeom geom;
        points_to_geom(pts, phase, &geom);
        return wide_backward(w, geom, N, g_raw, grads, work, work_bytes, (hipStream_t)stream);
    }
    static thread_local NcaFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = NCA_MODE_POINTS;
    a.nnets = 1;
    NetBind binds[2] = {{net, packed, win, four, params}, {}};
    rc = fill_net(binds[0], &a.net[0], prec);
    if (rc) return rc;
    if (a.net[0].lay.T > 0 && !phase) return fail(NCA_E_INVALID, "dynamic net needs phase ids");
    if (g_latents && a.net[0].lay.T <= 0) return fail(NCA_E_INVALID, "per-point latent gradients of a net without latents");
    if (g_latents)       // (a skip layer reads [enc, latents, hidden] again: its W[:, latent columns]^T D term is not formed here)
        for (int j = 0; j < a.net[0].lay.NL; ++j)
            if (a.net[0].lay.layer[j].kind == NCA_IN_SKIP) return fail(NCA_E_UNSUPPORTED, "per-point latent gradients of a net with a skip layer (the table-row sums in `grads` are complete)");
    a.N = N;
    a.pts = pts;
    a.phase = phase;
    a.g_raw = g_raw;
    float* gr[2] = {grads, nullptr};
    const int ts = tile_samples(prec);
    return run_bwd(a, prec, binds, (N + ts - 1) / ts, 1, gr, work, work_bytes, (hipStream_t)stream, nullptr, 0, nullptr, NCA_STORE_NONE, nullptr, g_latents);
}

// ---------------------------------------------------------------------------------- losses
extern "C" int64_t nca_loss_workspace(int64_t R) {
    if (R <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    return align_up(nca_loss_partials_bytes(R), 256);
}

extern "C" int nca_loss_fwd_bwd(const NcaLoss* d, const double* pix, const double* gt, const double* wpix,
                                const float* sig_s, const float* sig_d, const double* dists,
                                double* terms, double* g_pix, float* g_sig_s, float* g_sig_d,
                                void* work, int64_t work_bytes, void* stream) {
    if (!d) return fail(NCA_E_INVALID, "loss descriptor is NULL");
    if (d->R <= 0 || d->S <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    const bool tgm = d->term_grads != nullptr;       // term-gradient mode: no pixel term -- pix / gt / g_pix may be NULL
    if (d->ray_part) {           // pix is formed by the kernel from the forward's per-tile ray sums
        if (tgm) return fail(NCA_E_INVALID, "ray_part and term_grads exclude each other");
        if (!d->ray_I0 || d->ray_nchunk <= 0 || !gt) return fail(NCA_E_INVALID, "ray_part needs ray_I0, ray_nchunk > 0 and gt");
        pix = d->pix_out ? d->pix_out : reinterpret_cast<const double*>(d->ray_part);          // (non-null marker: the kernel reads ray_part, never this)
    }
    if ((!tgm && (!pix || !gt)) || !wpix || !sig_s || !sig_d || !dists || !terms) return fail(NCA_E_INVALID, "a loss input pointer is NULL");
    if ((pix == nullptr) != (gt == nullptr)) return fail(NCA_E_INVALID, "pix and gt: both or neither");
    if (tgm && d->weights_dev) return fail(NCA_E_INVALID, "term_grads and weights_dev exclude each other");
    const bool any = g_pix || g_sig_s || g_sig_d;
    if (any && !((g_pix || !pix) && g_sig_s && g_sig_d)) return fail(NCA_E_INVALID, "give all three gradient outputs or none (g_pix may be NULL where pix is)");
    if (g_pix && !pix) return fail(NCA_E_INVALID, "g_pix without pix");
    const int64_t need = nca_loss_workspace(d->R);
    if (!work || work_bytes < need) return fail(NCA_E_WORKSPACE, "loss workspace %lld < %lld bytes", (long long)work_bytes, (long long)need);
    NcaLossArgs a;
    a.R = d->R; a.S = d->S; a.use_weighting = d->use_weighting;
    a.skew = d->skew; a.mask_thre = d->mask_thre; a.weighted_thresh = d->weighted_thresh;
    a.w_favor = d->w_favor; a.w_dent = d->w_dent; a.w_occl = d->w_occl; a.w_l1 = d->w_l1; a.inv_R = d->inv_R;
    a.weights_dev = d->weights_dev;
    a.unit_mse = d->unit_mse;
    a.pix = pix; a.gt = gt; a.wpix = wpix; a.sig_s = sig_s; a.sig_d = sig_d; a.dists = dists;
    a.terms = terms; a.g_pix = g_pix; a.g_sig_s = g_sig_s; a.g_sig_d = g_sig_d;
    a.partials = static_cast<double*>(work);
    if (d->g_dists && !(any && d->dists_work)) return fail(NCA_E_INVALID, "g_dists needs the three gradient outputs and dists_work (f64[R * S])");
    a.g_dists = d->g_dists;
    a.dists_work = d->g_dists ? d->dists_work : nullptr;
    a.term_grads = d->term_grads;
    a.ray_part = d->ray_part; a.ray_I0 = d->ray_I0; a.pix_out = d->pix_out; a.ray_nchunk = d->ray_nchunk; a.pad2_ = 0;
    a.terms_f32 = d->terms_f32;
    if (a.ray_part) a.pix = nullptr;
    Span sp(NCA_K_LOSS, (hipStream_t)stream);
    HIPCHK(nca_launch_loss(a, (hipStream_t)stream));
    return NCA_OK;
}

extern "C" int nca_weighted_sq_err(int64_t R, int32_t is_f64, const void* pred, const void* gt, const void* w, void* out, void* stream) {
    if (R <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    if (!pred || !gt || !w || !out) return fail(NCA_E_INVALID, "a pointer is NULL");
    Span sp(NCA_K_LOSS, (hipStream_t)stream);
    HIPCHK(nca_launch_wsqerr(R, is_f64 != 0, pred, gt, w, out, (hipStream_t)stream));
    return NCA_OK;
}
extern "C" int nca_weighted_sq_err_bwd(int64_t R, int32_t is_f64, const void* pred, const void* gt, const void* w, const void* g_out,
                                       void* g_pred, void* g_gt, void* g_w, void* stream) {
    if (R <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    if (!pred || !gt || !w || !g_out) return fail(NCA_E_INVALID, "a pointer is NULL");
    Span sp(NCA_K_LOSS, (hipStream_t)stream);
    HIPCHK(nca_launch_wsqerr_bwd(R, is_f64 != 0, pred, gt, w, g_out, g_pred, g_gt, g_w, (hipStream_t)stream));
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- fine-pass depths
extern "C" int64_t nca_fine_depths_workspace(int64_t R) {
    if (R <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    return align_up((nca_fine_partials(R) + 1) * (int64_t)sizeof(float), 256);
}

static int fine_check(int64_t R, int32_t S, int32_t n_fine) {
    if (R <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    if (S < 3) return fail(NCA_E_INVALID, "fine sampling needs at least 3 coarse samples per ray (got %d)", S);
    if (n_fine < 1) return fail(NCA_E_INVALID, "n_fine must be positive");
    int npad = 64;
    while (npad < n_fine) npad <<= 1;
    if (4 * (int64_t)((S - 1) + npad + S) * 4 > 160 * 1024) return fail(NCA_E_UNSUPPORTED, "S + n_fine too large for the LDS-resident sampler");
    return NCA_OK;
}

extern "C" int nca_fine_weight_max(int64_t R, int32_t S, const float* sig_s, const float* sig_d, float* wmax, void* work, int64_t work_bytes,
                                   void* stream) {
    const int rc = fine_check(R, S, 1);
    if (rc != NCA_OK) return rc;
    if (!sig_s || !wmax) return fail(NCA_E_INVALID, "a pointer is NULL");
    const int64_t need = nca_fine_depths_workspace(R);
    if (!work || work_bytes < need) return fail(NCA_E_WORKSPACE, "fine-depth workspace %lld < %lld bytes", (long long)work_bytes, (long long)need);
    NcaFineArgs a{};
    a.R = R; a.S = S; a.n_fine = 1;
    a.sig_s = sig_s; a.sig_d = sig_d;
    a.partial_max = static_cast<float*>(work);
    a.jmax = wmax;
    HIPCHK(nca_launch_fine_max(a, (hipStream_t)stream));
    return NCA_OK;
}

extern "C" int nca_fine_depths_given_max(int64_t R, int32_t S, int32_t n_fine, const float* sig_s, const float* sig_d, const float* z,
                                         const float* u, const float* wmax, float* z_all, void* stream) {
    const int rc = fine_check(R, S, n_fine);
    if (rc != NCA_OK) return rc;
    if (!sig_s || !z || !u || !z_all || !wmax) return fail(NCA_E_INVALID, "a pointer is NULL");
    NcaFineArgs a{};
    a.R = R; a.S = S; a.n_fine = n_fine;
    a.sig_s = sig_s; a.sig_d = sig_d; a.z = z; a.u = u; a.z_all = z_all;
    a.jmax = const_cast<float*>(wmax);
    HIPCHK(nca_launch_fine_sample(a, (hipStream_t)stream));
    return NCA_OK;
}

extern "C" int nca_fine_depths(int64_t R, int32_t S, int32_t n_fine, const float* sig_s, const float* sig_d, const float* z,
                               const float* u, float* z_all, void* work, int64_t work_bytes, void* stream) {
    const int rc = fine_check(R, S, n_fine);
    if (rc != NCA_OK) return rc;
    if (!sig_s || !z || !u || !z_all) return fail(NCA_E_INVALID, "a pointer is NULL");
    const int64_t need = nca_fine_depths_workspace(R);
    if (!work || work_bytes < need) return fail(NCA_E_WORKSPACE, "fine-depth workspace %lld < %lld bytes", (long long)work_bytes, (long long)need);
    NcaFineArgs a{};
    a.R = R; a.S = S; a.n_fine = n_fine;
    a.sig_s = sig_s; a.sig_d = sig_d; a.z = z; a.u = u; a.z_all = z_all;
    a.partial_max = static_cast<float*>(work);
    a.jmax = a.partial_max + nca_fine_partials(R);
    HIPCHK(nca_launch_fine(a, (hipStream_t)stream));
    return NCA_OK;
}

// Backward of nca_fine_depths (the reference keeps the sampled depths in its autograd graph, model_helpers.py:135-146).
// Stage 1: g_tot f32[R,S] = d loss / d (sigma_s + sigma_d) through the sampling with the maximum held fixed, plus per-ray
// parts of d loss / d wmax and of the number of jumps that attain wmax.  The caller sums them (over the ranks too), divides,
// and stage 2 adds the quotient to the jumps that attain the maximum.
extern "C" int nca_fine_depths_bwd(int64_t R, int32_t S, int32_t n_fine, const float* sig_s, const float* sig_d, const float* z,
                                   const float* u, const float* wmax, const float* g_z_all, float* g_tot, float* gmax_part, float* cnt_part,
                                   void* stream) {
    const int rc = fine_check(R, S, n_fine);
    if (rc != NCA_OK) return rc;
    if (!sig_s || !z || !u || !wmax || !g_z_all || !g_tot || !gmax_part || !cnt_part) return fail(NCA_E_INVALID, "a pointer is NULL");
    if (4 * (int64_t)(2 * (S - 1) + 2 * S + 5 * n_fine) * 4 > 160 * 1024) return fail(NCA_E_UNSUPPORTED, "S + n_fine too large for the LDS-resident sampler backward");
    NcaFineBwdArgs a{};
    a.R = R; a.S = S; a.n_fine = n_fine;
    a.sig_s = sig_s; a.sig_d = sig_d; a.z = z; a.u = u; a.jmax = wmax; a.g_zall = g_z_all;
    a.g_tot = g_tot; a.gmax_part = gmax_part; a.cnt_part = cnt_part;
    HIPCHK(nca_launch_fine_bwd(a, (hipStream_t)stream));
    return NCA_OK;
}
extern "C" int nca_fine_depths_bwd_max(int64_t R, int32_t S, const float* sig_s, const float* sig_d, const float* wmax, const float* gmax_each,
                                       float* g_tot, void* stream) {
    const int rc = fine_check(R, S, 1);
    if (rc != NCA_OK) return rc;
    if (!sig_s || !wmax || !gmax_each || !g_tot) return fail(NCA_E_INVALID, "a pointer is NULL");
    NcaFineBwdArgs a{};
    a.R = R; a.S = S; a.n_fine = 1;
    a.sig_s = sig_s; a.sig_d = sig_d; a.jmax = wmax; a.gmax_each = gmax_each; a.g_tot = g_tot;
    HIPCHK(nca_launch_fine_bwd_max(a, (hipStream_t)stream));
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- batch preparation
extern "C" int nca_prepare_batch(int64_t R, int32_t S, const int64_t* ids, const double* table, const int64_t* phases, int64_t n_rows, int32_t* bad_ids,
                                 const float* depth, const float* t_rand,
                                 double* o, double* d, double* gt, double* w, int32_t* ph, float* z, double* dists, void* stream) {
    if (R <= 0 || S <= 0) return fail(NCA_E_INVALID, "empty ray batch (R=%lld, S=%d)", (long long)R, S);
    if (!ids || !table || !phases || !depth || !t_rand || !o || !d || !gt || !w || !ph || !z || !dists) return fail(NCA_E_INVALID, "a pointer is NULL");
    HIPCHK(nca_launch_prepare_batch(R, S, ids, table, phases, n_rows, bad_ids, depth, t_rand, o, d, gt, w, ph, z, dists, (hipStream_t)stream));
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- device-side sampling and schedules
static int check_sampler(const NcaSampler* s) {
    if (!s) return fail(NCA_E_INVALID, "sampler is NULL");
    if (s->R_global <= 0) return fail(NCA_E_INVALID, "sampler: R_global = %lld", (long long)s->R_global);
    if (s->n_var < 0 || s->n_var > s->R_global) return fail(NCA_E_INVALID, "sampler: n_var = %lld outside [0, R_global]", (long long)s->n_var);
    if (s->n_var > 0 && s->n_var_ids > 0) {
        if (!s->var_ids || !s->non_var_ids || s->n_non_var_ids <= 0)
            return fail(NCA_E_INVALID, "sampler: importance sampling needs both id tables (var %lld, non-var %lld ids)", (long long)s->n_var_ids, (long long)s->n_non_var_ids);
    } else if (s->n_rows <= 0) return fail(NCA_E_INVALID, "sampler: uniform sampling needs n_rows > 0");
    return NCA_OK;
}
extern "C" int nca_draw_ray_ids(const NcaSampler* s, int64_t slot0, int64_t R, int64_t* ids, void* stream) {
    int rc = check_sampler(s);
    if (rc) return rc;
    if (R <= 0 || slot0 < 0 || slot0 + R > s->R_global) return fail(NCA_E_INVALID, "slots [%lld, %lld) outside the global batch of %lld", (long long)slot0, (long long)(slot0 + R), (long long)s->R_global);
    if (!ids) return fail(NCA_E_INVALID, "ids is NULL");
    HIPCHK(nca_launch_draw_ray_ids(*s, slot0, R, ids, (hipStream_t)stream));
    return NCA_OK;
}
extern "C" int nca_draw_uniform(const NcaSampler* s, int32_t stream_id, int64_t n, float* out, void* stream) {
    if (!s || !out) return fail(NCA_E_INVALID, "a pointer is NULL");
    if (n <= 0 || stream_id < 0 || stream_id > 255) return fail(NCA_E_INVALID, "n = %lld, stream %d (0 .. 255)", (long long)n, stream_id);
    HIPCHK(nca_launch_draw_uniform(*s, stream_id, n, out, (hipStream_t)stream));
    return NCA_OK;
}
extern "C" int nca_begin_step(const NcaSampler* s, int64_t slot0, int64_t R, int32_t S, const NcaSchedules* sched,
                              const int64_t* ids_in, const float* t_rand_in,
                              const double* table, const int64_t* phases, int32_t* bad_ids, const float* depth,
                              double* o, double* d, double* gt, double* w, int32_t* ph, float* z, double* dists,
                              int64_t* ids_out, float* t_rand_out, void* stream) {
    if (!s) return fail(NCA_E_INVALID, "sampler is NULL");
    if (R <= 0 || S <= 0) return fail(NCA_E_INVALID, "empty ray batch (R=%lld, S=%d)", (long long)R, S);
    if (!ids_in) {
        int rc = check_sampler(s);
        if (rc) return rc;
        if (slot0 < 0 || slot0 + R > s->R_global) return fail(NCA_E_INVALID, "slots [%lld, %lld) outside the global batch of %lld", (long long)slot0, (long long)(slot0 + R), (long long)s->R_global);
    }
    if (!table || !phases || !depth || !o || !d || !gt || !w || !ph || !z || !dists) return fail(NCA_E_INVALID, "a pointer is NULL");
    NcaBeginArgs a;
    memset(&a, 0, sizeof(a));
    a.s = *s; a.slot0 = slot0; a.R = R; a.S = S;
    if (sched) {
        if (sched->n_windows < 0 || sched->n_windows > 4) return fail(NCA_E_INVALID, "n_windows = %d", sched->n_windows);
        for (int i = 0; i < sched->n_windows; ++i) {
            const NcaWindowSched& ws = sched->window[i];
            if (ws.kind != NCA_WINDOW_NONE && ws.kind != NCA_WINDOW_FREE) return fail(NCA_E_UNSUPPORTED, "window schedule %d: kind %d", i, ws.kind);
            if (ws.kind == NCA_WINDOW_FREE && (ws.L < 1 || ws.L > 64 || ws.decay_steps <= 0 || !ws.out)) return fail(NCA_E_INVALID, "window schedule %d: L = %d, decay_steps = %lld", i, ws.L, (long long)ws.decay_steps);
        }
        if (sched->weights_out)
            for (int i = 0; i < 4; ++i) if (sched->weight[i].steps <= 0) return fail(NCA_E_INVALID, "weight schedule %d: steps = %lld", i, (long long)sched->weight[i].steps);
        a.sch = *sched;
    }
    a.ids_in = ids_in; a.t_rand_in = t_rand_in; a.table = table; a.phases = phases; a.bad_ids = bad_ids; a.depth = depth;
    a.o = o; a.d = d; a.gt = gt; a.w = w; a.ph = ph; a.z = z; a.dists = dists; a.ids_out = ids_out; a.t_rand_out = t_rand_out;
    HIPCHK(nca_launch_begin_step(a, (hipStream_t)stream));
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- optimiser
extern "C" int nca_adam_step(const NcaAdam* cfg, int32_t n_seg, const int64_t* n, float* const* params, const float* const* grads,
                             float* const* exp_avg, float* const* exp_avg_sq, int64_t* step, void* stream) {
    if (!cfg || !n || !params || !grads || !exp_avg || !exp_avg_sq || !step) return fail(NCA_E_INVALID, "an Adam argument is NULL");
    if (n_seg < 1 || n_seg > NCA_ADAM_MAX_SEG) return fail(NCA_E_INVALID, "n_seg %d outside 1..%d", n_seg, (int)NCA_ADAM_MAX_SEG);
    if (!(cfg->lr >= 0.0) || !(cfg->beta1 >= 0.0 && cfg->beta1 < 1.0) || !(cfg->beta2 >= 0.0 && cfg->beta2 < 1.0) || !(cfg->eps >= 0.0))
        return fail(NCA_E_INVALID, "Adam hyper-parameters out of range");
    NcaAdamArgs a{};
    a.lr = cfg->lr; a.beta1 = cfg->beta1; a.beta2 = cfg->beta2; a.eps = cfg->eps;
    a.lr_end_factor = cfg->lr_end_factor; a.lr_total_iters = cfg->lr_total_iters;
    a.n_seg = n_seg; a.step = step; a.iter_counter = cfg->iter_counter;
    for (int s = 0; s < n_seg; ++s) {
        if (n[s] <= 0 || !params[s] || !grads[s] || !exp_avg[s] || !exp_avg_sq[s]) return fail(NCA_E_INVALID, "Adam segment %d is empty or NULL", s);
        a.n[s] = n[s]; a.params[s] = params[s]; a.grads[s] = grads[s]; a.exp_avg[s] = exp_avg[s]; a.exp_avg_sq[s] = exp_avg_sq[s];
    }
    Span sp(NCA_K_ADAM, (hipStream_t)stream);
    HIPCHK(nca_launch_adam(a, (hipStream_t)stream));
    return NCA_OK;
}

// ---------------------------------------------------------------------------------- stand-alone compositing
extern "C" int nca_composite_fwd(int64_t R, int32_t S, int32_t act, int32_t single_field, float scale,
                                 const float* raw_s, const float* raw_d, const float* I0, const double* dists,
                                 double* pix, float* sig_s, float* sig_d, void* stream) {
    if (R <= 0 || S <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    if (act < 0 || act > 2) return fail(NCA_E_INVALID, "unknown activation %d", act);
    if (!raw_s || !I0 || !dists || !pix || !sig_s || (!single_field && (!raw_d || !sig_d))) return fail(NCA_E_INVALID, "a pointer is NULL");
    NcaCompositeArgs a{};
    a.R = R; a.S = S; a.act = act; a.single = single_field; a.scale = scale;
    a.raw_s = raw_s; a.raw_d = raw_d; a.I0 = I0; a.dists = dists; a.pix = pix; a.sig_s = sig_s; a.sig_d = sig_d;
    HIPCHK(nca_launch_composite(a, false, (hipStream_t)stream));
    return NCA_OK;
}

extern "C" int nca_composite_bwd(int64_t R, int32_t S, int32_t act, int32_t single_field, float scale,
                                 const float* raw_s, const float* raw_d, const double* dists,
                                 const double* g_pix, const float* g_sig_s, const float* g_sig_d,
                                 float* g_raw_s, float* g_raw_d, void* stream) {
    if (R <= 0 || S <= 0) return fail(NCA_E_INVALID, "empty ray batch");
    if (act < 0 || act > 2) return fail(NCA_E_INVALID, "unknown activation %d", act);
    if (!raw_s || !dists || !g_raw_s || (!single_field && (!raw_d || !g_raw_d))) return fail(NCA_E_INVALID, "a pointer is NULL");
    NcaCompositeArgs a{};
    a.R = R; a.S = S; a.act = act; a.single = single_field; a.scale = scale;
    a.raw_s = raw_s; a.raw_d = raw_d; a.dists = dists;
    a.g_pix = g_pix; a.g_sig_s = g_sig_s; a.g_sig_d = g_sig_d; a.g_raw_s = g_raw_s; a.g_raw_d = g_raw_d;
    HIPCHK(nca_launch_composite(a, true, (hipStream_t)stream));
    return NCA_OK;
}
