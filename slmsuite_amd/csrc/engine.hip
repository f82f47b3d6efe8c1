// Host side of the hologram engine: device state, kernel sequencing, the C ABI of include/hgs.h.
// One engine = one HIP stream + all device buffers of a batch of equally-shaped holograms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hgs.h"
#include "launch.hpp"
#include "compressed_kernels.hpp"
#include "cgemm.hpp"
#include "compressed_sep.hpp"
#include "bluestein.hpp"
#include <complex>
#include <dlfcn.h>

namespace hgs {

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail(HGS_ERR_DEVICE, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define LCHK(x)                                                                                \
    do {                                                                                       \
        int e_ = (x);                                                                          \
        if (e_ != 0)                                                                           \
            return fail(HGS_ERR_DEVICE, "kernel launch failed: %s (%s:%d)",                    \
                        hipGetErrorString((hipError_t)e_), __FILE__, __LINE__);                \
    } while (0)

// roctx ranges around the operators (HGS_OPT_ROCTX): resolved at run time so that the library keeps its single
// link dependency (libamdhip64); rocprofv3 --marker-trace shows them.  librocprofiler-sdk-roctx first (rocprofv3),
// then the legacy libroctx64.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
    bool load() {
        if (tried) return push != nullptr;
        tried = true;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return true;
                push = nullptr;
                pop = nullptr;
            }
        }
        return false;
    }
};
static Roctx g_roctx;
struct RoctxRange {
    bool on;
    RoctxRange(bool enabled, const char* name) : on(enabled && g_roctx.push != nullptr) { if (on) g_roctx.push(name); }
    ~RoctxRange() { if (on) g_roctx.pop(); }
};

thread_local DispatchLog* g_dispatch = nullptr;
// one line per (kernel instance, run-time flags): "family<P0=v0,...> flag ...\tcount\n".  The argument values come from
// __PRETTY_FUNCTION__ of dispatch_site<Fam, R, V...>: "... [Fam = hgs::KTile, R = float, V = <4096, 1, 6, false, false, 1, 0>]"
std::string DispatchLog::text() const {
    static const char* flag_names[] = {"list", "load_mask", "store_mask", "xmap", "batch", "stats", "nf_out"};
    std::string out;
    for (const Ent& e : v) {
        std::vector<std::string> vals;
        const std::string p = e.site->pretty;
        size_t r0 = p.find("R = ");
        if (r0 != std::string::npos) {
            r0 += 4;
            const size_t r1 = p.find_first_of(",]", r0);
            vals.push_back(p.substr(r0, r1 - r0));
        }
        size_t v0 = p.find("V = <");
        if (v0 != std::string::npos) {
            v0 += 5;
            const size_t v1 = p.find('>', v0);
            std::string list = p.substr(v0, v1 - v0);
            size_t a = 0;
            while (a <= list.size()) {
                size_t b = list.find(", ", a);
                if (b == std::string::npos) b = list.size();
                if (b > a) vals.push_back(list.substr(a, b - a));
                a = b + 2;
            }
        }
        out += e.site->family;
        out += '<';
        const std::string names = e.site->params;
        size_t a = 0;
        for (size_t i = 0; i < vals.size(); ++i) {
            size_t b = names.find(',', a);
            if (b == std::string::npos) b = names.size();
            if (i) out += ',';
            out += (a < names.size() ? names.substr(a, b - a) : std::string("?")) + "=" + vals[i];
            a = b + 1;
        }
        out += '>';
        for (unsigned bit = 0; bit < sizeof flag_names / sizeof flag_names[0]; ++bit)
            if (e.flags & (1u << bit)) { out += ' '; out += flag_names[bit]; }
        out += '\t' + std::to_string(e.count) + '\n';
    }
    return out;
}

static bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct EngineBase {
    int device = 0;      // HIP device ordinal of this engine; every C-ABI entry makes it current first
    DispatchLog dispatch;    // kernel instances launched since the last hgs_dispatch_read
    virtual ~EngineBase() {}
    virtual int init(const hgs_config& c) = 0;
    virtual int set_array(int which, const void* src, size_t nbytes, bool src_device) = 0;
    // device-resident phase of another engine (same SLM shape, precision and batch); PhaseRef: see phase_ref()
    struct PhaseRef { const void* ptr; size_t S; int B; int real_bytes; int device; hipStream_t stream; };
    virtual int phase_ref(PhaseRef* out) = 0;
    virtual int copy_phase_from(const PhaseRef& src) = 0;
    virtual int get_array(int which, void* dst, size_t nbytes, bool dst_device) = 0;
    virtual int reset_weights() = 0;
    virtual int reset_state() = 0;
    virtual int set_array_sparse(int which, const int32_t* xy, const void* values, int n) = 0;
    virtual int n2f(int store_pff) = 0;
    virtual int constraint(hgs_step* st) = 0;
    virtual int f2n() = 0;
    // MultiplaneHologram support: inverse transform without phase extraction, and the cross-engine combine
    struct MpInfo { void* nf; const void* kern; void* phase; size_t S; int B; int real_bytes; int device; hipStream_t stream; };
    virtual int f2n_complex() = 0;
    virtual int mp_info(MpInfo* out) = 0;
    virtual int mp_combine(const MpInfo* infos, const double* weights, int n) = 0;
    virtual int iterate(hgs_step* st, int n, uint8_t* hist) = 0;
    virtual int iterate_stats(hgs_step* st, int n, uint8_t* hist, int groups, int width, const double* xy,
                              double* out) = 0;
    virtual int stats(int group, int width, const double* xy, double* out) = 0;
    virtual int sync() = 0;
    virtual int set_option(int option, int value) = 0;
    virtual int profile_enable(int on) = 0;
    virtual int profile_read(double* out) = 0;
    virtual int iterate_timed(hgs_step* st, int n, double* ms) = 0;
};

template <typename R> struct Engine : EngineBase {
    using C = Cx<R>;
    hgs_config cfg{};
    Geo g{};
    hipStream_t stream = nullptr;
    size_t S = 0, P = 0;  // elements per hologram
    int B = 1;
    // device buffers
    R* phase = nullptr;
    R* amp = nullptr;
    R* kern = nullptr;
    C* gh = nullptr;
    C* gh2 = nullptr;         // single-pass MRAF: the noise-region part of the field between the column and the row kernel
    R* w = nullptr;
    R* t = nullptr;
    R* pff = nullptr;
    C* ff = nullptr;
    R* aff = nullptr;
    C* zw = nullptr;
    void* staging = nullptr;  // B*P complex, natural-layout bounce buffer for set/get
    C* tw_row = nullptr;
    C* tw_col = nullptr;
    double* wpartial = nullptr;
    double* dpartial = nullptr;  // partials of col_presum_kernel (single-inverse MRAF), sized like wpartial
    double* fpartial = nullptr;
    double* epartial = nullptr;  // elementwise partials
    double* sums = nullptr;      // [3][B]: fsum, nogsum, wsum
    R* wscale = nullptr;
    int* spot_xy = nullptr;
    double* spot_amp = nullptr;
    double* ext_amp = nullptr;
    R* spot_fb = nullptr;
    C* nfbuf = nullptr;               // [B][Sh][Sw] complex nearfield of f2n_complex (MultiplaneHologram)
    R* nog_dev = nullptr;             // [B] -1/mean(fc) of the fused WGS-Nogrette pass
    // sparse targets (spot arrays): columns that hold a non-zero weight or target
    unsigned char* col_active = nullptr;   // [B][Pw]
    unsigned short* sig_rows = nullptr;    // [B][Pw] register slots of a column that hold signal pixels (scan_active_cols; col_presum_kernel)
    int* col_list = nullptr;               // [B][Pw] compacted
    int* n_active_dev = nullptr;           // [B]
    unsigned short* lane_mask = nullptr;   // [B][Pw/16] row-kernel view of col_active
    unsigned short* lane_mask_noise = nullptr;   // ... of its bit 4: columns with a NaN target (noise part of single-pass MRAF)
    int opt_gh2_mask = 1;                  // developer A/B (HGS_GH2_MASK=0 at create): noise part stored / read for every column
    // the same for the columns the spot integration windows touch (spot feedback / spot statistics)
    unsigned char* col_active_d = nullptr;
    int* col_list_d = nullptr;
    int* n_active_d_dev = nullptr;
    unsigned short* lane_mask_d = nullptr;
    int n_active_d_max = 0, dil_lo = 0, dil_hi = 0;
    bool dil_valid = false;
    int n_active_max = 0, n_active_min = 0;
    bool sparse_dirty = true;
    bool sparse_tiles = false;             // the active set is whole 4-column tiles (the tile-resident kernel walks the list)
    int opt_tile_list = 1;                 // developer A/B (HGS_TILE_LIST=0 at create): column lists always go to the per-column kernel
    // engine policy (hgs_set_option); the grid-size tuning knobs are read from the environment once, in init()
    int opt_sparse = 1;                    // HGS_OPT_SPARSE_COLUMNS
    int opt_stepwise = 0;                  // HGS_OPT_FORCE_STEPWISE
    int opt_tile = 1;                      // HGS_OPT_TILE_KERNEL
    int opt_separable = 1;                 // HGS_OPT_SEPARABLE
    int opt_sep_min = 96;                  // smallest spot count the matrix-core form is used for (tools/sep_crossover.py)
    int opt_roctx = 0;                     // HGS_OPT_ROCTX: roctx ranges around the operators
    int opt_tile_rule = 1;                 // developer A/B (HGS_TILE_RULE=0 at create): rule-specialised tile kernels off
    // G left behind (round 5): the last launch of a fused float32 hgs_iterate call is row_kernel MODE 3 -- it writes the phase AND
    // the row-transformed field of the next body, every column of it -- and the next call (or hgs_nearfield2farfield) skips its
    // own first row launch while nothing that G depends on (phase, amplitude, kernel) has changed.
    //   gh_state: -1 = gh does not hold G; 0 = G of every column; 1 = of the active columns; 2 = of the dilated active columns
    int gh_state = -1;
    // HGS_OPT_KEEP_PREV_PHASE: the phase a one-iteration fused call started from (what its farfield phase describes)
    R* phase_prev = nullptr;
    bool have_prev = false;
    int opt_prev_phase = 0;
    int opt_tile2 = 1;                     // developer A/B (HGS_TILE2=0 at create): half-width tile kernel off
    int opt_tile2_phase2 = 1;              // ... its phase-reading form at 4096 rows, one hologram (HGS_TILE2_PHASE2=0 at create: col_tile_kernel)
    int opt_tile2_min_batch = 1;           // ... smallest batch that runs it at 4096 rows (HGS_TILE2_MIN_BATCH)
    int env_tile2_blocks = 0;              // ... its workgroups per launch over the batch (HGS_TILE2_BLOCKS; 0 = 3 x / 2 x #CU)
    int opt_mono_tab = 1;                  // developer A/B (HGS_MONO_TAB=0 at create): per-pixel compressed kernels evaluate every monomial per spot
    int opt_keep_g = 1;                    // developer A/B (HGS_KEEP_G=0 at create)
    int opt_mraf_presum = 1;               // developer A/B (HGS_MRAF_PRESUM=0 at create: the two-inverse split form on every update)
    int env_presum_blocks = 0, opt_presum_rows = 1;    // developer A/B (HGS_PRESUM_BLOCKS, HGS_PRESUM_ROWS at create)
    int opt_fused_shift = 1;               // developer A/B (HGS_FUSED_SHIFT=0 at create): float64 per-column kernel unshifted (16 slots)
    int opt_tile_nr4 = 1;                  // developer A/B (HGS_TILE_NR4=0 at create): slot-count instances of the rule kernels off (NR = 6 only)
    int opt_tile_shift16 = 1;              // developer A/B (HGS_TILE_SHIFT16=0 at create): the tile kernel shifts by whole register slots
    int opt_row_shift = 1;                 // developer A/B (HGS_ROW_SHIFT=0 at create): shifted row kernel off
    int opt_row_shift64 = 1;               // ... in float64 (HGS_ROW_SHIFT64=0 at create; round 5)
    int opt_row_pref = 1;                  // developer A/B (HGS_ROW_PREF=0 at create): prefetching row kernel off
    int opt_mraf_split = 1;                // developer A/B (HGS_MRAF_SPLIT=0 at create): MRAF weight updates in two column passes
    bool row_split = false;                // the next row kernel joins gh and gh2 (single-pass MRAF)
    // per-column kernel (float64; float32 where the tile-resident kernel does not run) single-pass MRAF: noise part as farfield
    // values, the columns that hold it, their inverse pass
    C* ffb = nullptr;                      // [B][P], layout of ff; only NaN-target pixels are ever written, the rest stays zero
    int* col_list_noise = nullptr;         // [B][Pw] columns with a NaN target (bit 4 of col_active), compacted
    int* n_noise_dev = nullptr;            // [B]
    unsigned short* lane_mask_tmp = nullptr;
    int n_noise_max = 0;
    bool noise_valid = false;              // col_list_noise matches the current target
    int* col_list_signal = nullptr;        // [B][Pw] columns with a finite non-zero target (bit 1 of col_active), compacted: the
    int* n_signal_dev = nullptr;           // [B]     per-column pre-pass of the single-inverse MRAF update walks them
    int n_signal_max = 0;
    bool signal_valid = false;
    bool ffb_zeroed = false;               // ... and so do the zeros of ffb (written since at NaN-target pixels only)
    bool row_split_noise_only = false;     // ... and the row kernel must read gh2 in those columns only (nothing else was written)
    int opt_mraf_split64 = 1;              // developer A/B (HGS_MRAF_SPLIT64=0 at create): float64 MRAF weight updates in two passes
    int row_blocks_pref = 0;               // its grid: two workgroups per CU, whole XCD line groups
    // statistics of the fused path (hgs_iterate_stats)
    double* stats_scratch = nullptr;  // hgs_stats group 0: per-block partials of the two passes
    int* stats_dxy = nullptr;         // hgs_stats group 1: floor(spot_knm)
    double* stat_partial = nullptr;   // [B][blocks][STAT_WAVES][STAT_N]
    double* stat_tsum = nullptr;      // [B] sum T^2
    struct StatCtx { int groups = 0, width = 1; double* dev_out = nullptr; int* dxy = nullptr; };
    StatCtx* stat_ctx = nullptr;      // non-null while hgs_iterate_stats drives the fused loop
    size_t stat_nslots = 0;
    // padded shapes that are not powers of two in [64, 8192]: Bluestein path (bluestein.hpp)
    bool general = false;
    int blue_M[2] = {0, 0};            // convolution lengths for x (rows, N = Pw) and y (columns, N = Ph)
    bool blue_plain[2] = {false, false};   // the axis is a power of two: M = N, no convolution
    C* blue_tab[2][2][3] = {};         // [x|y][forward|inverse][A, Bf, Cc]
    C* blue_tw[2] = {nullptr, nullptr};
    // kind 1 (compressed)
    R* xg = nullptr;
    R* yg = nullptr;
    int* mono = nullptr;
    R* coeff = nullptr;
    Cx<R>* cpartial = nullptr;
    double* cnorm = nullptr;
    R* ext_r = nullptr;
    int c_nblocks = 0, c_degree = -1, c_rows = 0;
    // separable (matrix-core) form of the compressed transforms, fp32 only (compressed_sep.hpp)
    bool grid_sep[2] = {false, false}, c_sep = false;
    std::vector<double> xs_host, ys_host;
    double* sep_c = nullptr;        // [2][SEP_MAXDEG+1][N] polynomial coefficients of fx_n, fy_n
    double* sep_g = nullptr;        // xs[W] then ys[H]
    float2* sep_ex = nullptr;       // [N][W]
    float2* sep_exT = nullptr;      // [W][N]
    float2* sep_ey = nullptr;       // [N][H]
    float2* sep_nfT = nullptr;      // [B][W][H]
    float2* sep_b2 = nullptr;       // [B][N][H]
    float2* sep_c1 = nullptr;       // [B][tiles_n1 * 2 * split1][Np] partial y contractions of the n2f GEMM epilogue
    float2* sep_c2 = nullptr;       // [B][split2][H][W]
    double* sep_norm = nullptr;     // [B][ceil(N/4)]
    int sep_split1 = 1, sep_kper1 = 0, sep_split2 = 1, sep_kper2 = 0, sep_degx = 0, sep_degy = 0;
    // stream-K schedule of the two GEMMs (cgemm_streamk): sep_split1 / sep_split2 are then the partial planes of C
    int sk_G1 = 0, sk_G2 = 0, sk_kt1 = 0, sk_kt2 = 0, sk_tm1 = 0, sk_tn1 = 0, sk_tm2 = 0, sk_tn2 = 0;
    int* sk_tab = nullptr;          // [first_wg 1][nseg 1][first_wg 2][nseg 2]
    int sep_Np = 0, sep_Hp = 0, sep_Wp = 0, sep_Wk = 0, sep_Nk = 0;   // padded leading dimensions / row counts
    std::vector<int32_t> mono_host;
    std::vector<R> coeff_host;
    bool has_grid[2] = {false, false}, has_mono = false, has_coeff = false;
    // run kernels of the direct compressed transforms (compressed_kernels.hpp: regular grid, degree <= 2, fp32)
    bool run_ok = false;
    int opt_run = 1;                       // HGS_OPT_RUN_KERNELS
    CRunRec* run_rec = nullptr;            // [N]
    double* run_ys = nullptr;              // [H]
    Cx<float>* run_nf = nullptr;           // [B][run_chunks][S]
    double run_x0 = 0, run_hx = 0;
    int run_rpr = 0, run_blocks = 0, run_chunks = 1, run_nper = 0, run_nf_chunks = 0;
    // host state
    double amp_scalar = 0, amp_norm2 = 1.0;
    bool has_amp = false, has_kern = false, have_pff = false, farfield_valid = false;
    bool w_pending = false;  // weights stored un-normalised, wscale holds 1/||w||
    // ... and wscale^2 * sum w^2 = 1 to rounding: the stored weights were last written by an update pass of the fused loop and
    // wscale was folded from THAT pass' partial sums (no NaN left among them).  What the single-inverse MRAF pass builds on
    // (||w'||^2 = 1 + D); every other writer of the weights or of wscale goes through fill_wscale_one and clears it.
    bool w_unit = false;
    bool has_target = false, has_spots = false;
    int row_blocks = 0, col_blocks = 0, ew_blocks = 0, n_cu = 256, row_xcd = 0, tile_blocks = 0, wpartial_n = 0, col_xmap = 0, list_xmap = 0;
    // profiling
    bool prof = false;
    struct Ev { int kind; hipEvent_t a, b; };
    std::vector<Ev> evs;
    double prof_ms[HGS_K_COUNT] = {0};
    double prof_n[HGS_K_COUNT] = {0};

    ~Engine() override {
        if (stream) hipStreamSynchronize(stream);
        if (tw_col == tw_row) tw_col = nullptr;
        void* ptrs[] = {phase_prev, ffb, col_list_signal, n_signal_dev, col_list_noise, n_noise_dev, lane_mask_tmp, lane_mask_noise, phase, amp, kern, gh, gh2, w, t, pff, ff, aff, zw, staging, tw_row, tw_col, wpartial, dpartial,
                        fpartial, epartial, sums, wscale, spot_xy, spot_amp, ext_amp, spot_fb, nfbuf, nog_dev, stats_scratch, stats_dxy, col_active, sig_rows, col_list, n_active_dev, lane_mask, col_active_d, col_list_d, n_active_d_dev, lane_mask_d, stat_partial, stat_tsum, xg, yg, mono, coeff, cpartial, cnorm, ext_r, sep_c, sep_g, sep_ex, sep_exT, sep_ey, sep_nfT, sep_b2, sep_c1, sep_c2, sep_norm, run_rec, run_ys, run_nf, sk_tab};
        for (void* p : ptrs)
            if (p) hipFree(p);
        for (auto& d : blue_tab) for (auto& dir : d) for (C* t3 : dir) if (t3) hipFree(t3);
        if (blue_tw[0]) hipFree(blue_tw[0]);
        if (blue_tw[1] && blue_tw[1] != blue_tw[0]) hipFree(blue_tw[1]);
        for (auto& e : evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        for (hipEvent_t ev : timed_ev) if (ev) hipEventDestroy(ev);
        if (stream) hipStreamDestroy(stream);
    }

    // host -> device on the engine stream, complete on return (ordered against in-flight kernels of this
    // engine; the caller's buffer may be reused immediately)
    int h2d(void* dst, const void* src, size_t nbytes, bool src_device = false) {
        HIPCHK(hipMemcpyAsync(dst, src, nbytes, src_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    template <typename T> int dalloc(T** p, size_t n) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
        HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(T), stream));
        return 0;
    }

    int make_twiddles(C** dev, int N) {
        std::vector<C> h(N);
        for (int i = 0; i < N; ++i) {
            const double a = -2.0 * M_PI * (double)i / (double)N;
            h[i].x = (R)cos(a);
            h[i].y = (R)sin(a);
        }
        // exact values on the axes (cos(pi/2) is 6e-17 in double, rounds fine, but keep them exact)
        h[0].x = 1; h[0].y = 0;
        h[N / 4].x = 0; h[N / 4].y = -1;
        h[N / 2].x = -1; h[N / 2].y = 0;
        h[3 * N / 4].x = 0; h[3 * N / 4].y = 1;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(dev), N * sizeof(C)));
        if (int e_ = h2d(*dev, h.data(), N * sizeof(C))) return e_;
        return 0;
    }

    int init(const hgs_config& c) override {
        cfg = c;
        device = c.device;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(HGS_ERR_DEVICE, "no HIP device available (the engine has no CPU fallback)");
        if (c.device < 0 || c.device >= ndev) return fail(HGS_ERR_ARG, "device %d out of range", c.device);
        HIPCHK(hipSetDevice(c.device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, c.device));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (c.kind == 1) return init_compressed(c);
        if (c.kind != 0) return fail(HGS_ERR_ARG, "unknown engine kind %d", c.kind);
        const bool fast = is_pow2(c.pad_h) && is_pow2(c.pad_w) && c.pad_h >= 64 && c.pad_w >= 64 && c.pad_h <= 8192 &&
                          c.pad_w <= 8192;
        if (!fast) {
            // any other shape runs axis by axis on the workgroup transforms: a power-of-two axis up to 16384 (fp64:
            // 8192) directly, any other length up to 8192 (fp64: 4096) through Bluestein's identity
            if (!axis_ok(c.pad_h) || !axis_ok(c.pad_w))
                return fail(HGS_ERR_UNSUPPORTED, "padded shape (%d, %d): per axis a power of two up to %d or any length "
                            "in [2, %d] (%s)", c.pad_h, c.pad_w, max_line(), max_line() / 2, sizeof(R) == 4 ? "float32" : "float64");
            general = true;
        }
        if (c.slm_h < 1 || c.slm_w < 1 || c.slm_h > c.pad_h || c.slm_w > c.pad_w)
            return fail(HGS_ERR_ARG, "slm shape (%d, %d) does not fit the padded shape (%d, %d)", c.slm_h,
                        c.slm_w, c.pad_h, c.pad_w);
        if (c.batch < 1) return fail(HGS_ERR_ARG, "batch must be >= 1");
        g.Ph = c.pad_h; g.Pw = c.pad_w; g.Sh = c.slm_h; g.Sw = c.slm_w;
        g.r0 = (c.pad_h - c.slm_h) / 2;  // floor((P-S)/2)  toolbox.unpad
        g.c0 = (c.pad_w - c.slm_w) / 2;
        g.batch = B = c.batch;
        g.lane_T = general ? 0 : g.Ph / 16;
        S = (size_t)g.Sh * g.Sw;
        P = (size_t)g.Ph * g.Pw;
        // HGS_TRACE_INIT=1: where hgs_create spends its time (developer aid, stderr)
        const bool trace_init = env_int("HGS_TRACE_INIT", 0) != 0;
        opt_tile_rule = env_int("HGS_TILE_RULE", 1);
        opt_tile_nr4 = env_int("HGS_TILE_NR4", 1);
        opt_fused_shift = env_int("HGS_FUSED_SHIFT", 1);
        opt_keep_g = env_int("HGS_KEEP_G", 1);
        opt_tile2 = env_int("HGS_TILE2", 1);
        opt_tile2_min_batch = env_int("HGS_TILE2_MIN_BATCH", 1);
        opt_tile2_phase2 = env_int("HGS_TILE2_PHASE2", 1);
        env_tile2_blocks = env_int("HGS_TILE2_BLOCKS", 0);
        opt_tile_shift16 = env_int("HGS_TILE_SHIFT16", 1);
        opt_row_pref = env_int("HGS_ROW_PREF", 1);
        opt_mraf_split = env_int("HGS_MRAF_SPLIT", 1);
        opt_mraf_split64 = env_int("HGS_MRAF_SPLIT64", 1);
        opt_mraf_presum = env_int("HGS_MRAF_PRESUM", 1);
        opt_presum_rows = env_int("HGS_PRESUM_ROWS", 1);
        env_presum_blocks = env_int("HGS_PRESUM_BLOCKS", 0);
        opt_gh2_mask = env_int("HGS_GH2_MASK", 1);
        opt_tile_list = env_int("HGS_TILE_LIST", 1);
        opt_row_shift = env_int("HGS_ROW_SHIFT", 1);
        opt_row_shift64 = env_int("HGS_ROW_SHIFT64", 1);
        auto t_prev = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!trace_init) return;
            const auto t_now = std::chrono::steady_clock::now();
            fprintf(stderr, "hgs_create: %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t_now - t_prev).count());
            t_prev = t_now;
        };
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        lap("stream");
        if (general) return init_general(c);

        // grid sizes: enough workgroups to fill the chip a few times over, balanced over the work
        const int fpw = row_fpw(g.Pw);
        const int row_units = (g.Sh + fpw - 1) / fpw;
        // one-row workgroups balance best (the hardware dispatcher hands a free slot the next row); a batch keeps them up
        // to 64 workgroups per CU (cfg 3, 8 x 1152 rows: 187.6 -> 171.4 us per row launch against 256 per hologram)
        int cap = env_int("HGS_ROW_BLOCKS", n_cu * 64);
        cap = cap / B > 0 ? cap / B : 1;
        int per = (row_units + cap - 1) / cap;
        row_blocks = (row_units + per - 1) / per;
        // XCD-aware row mapping (row_kernel): the 4 / fpw workgroups of a 128-byte line group on one XCD together
        const int grp = fpw <= 4 ? 4 / fpw : 1;
        row_xcd = (grp > 1 && row_blocks >= 8 * grp && env_int("HGS_ROW_XCD", 1)) ? 1 : 0;
        if (row_xcd) row_blocks = (row_blocks + 8 * grp - 1) / (8 * grp) * (8 * grp);
        // prefetching row kernel (one hologram; a batch keeps the one-row workgroups): 2 workgroups per CU
        // (round 5: a batch walks too where every workgroup gets at least four rows -- 8 x 1152 rows over 2 x #CU workgroups
        //  are 18 rows each, no partial round at all; HGS_ROW_PREF_BATCH=0 keeps its one-row workgroups)
        row_blocks_pref = 0;
        if (sizeof(R) == 4 && g.Pw == 4096 && fpw == 1 && row_xcd && (B == 1 || env_int("HGS_ROW_PREF_BATCH", 0))) {
            const int want = env_int("HGS_ROW_PREF_BLOCKS", 2 * n_cu) / B;
            row_blocks_pref = std::max(8 * grp, want / (8 * grp) * (8 * grp));
        }
        const int tiles = g.Pw / 4;
        cap = env_int("HGS_COL_BLOCKS", n_cu * 3);
        cap = cap / B > 0 ? cap / B : 1;
        per = (tiles + cap - 1) / cap;
        col_blocks = (tiles + per - 1) / per;
        // per-column fused kernel with several passes per tile (ColCfg: fewer than four columns side by side): a grid that
        // is a multiple of 8 * passes lets the passes of a tile run on one XCD at a time (ColArgs::col_xmap); among those
        // the one that wastes the least of its last sweep, the larger on a tie
        col_xmap = 0;
        list_xmap = env_int("HGS_COL_XMAP", 1);
        {
            const int Tc = g.Ph / 16, cpar = Tc >= 256 ? 1 : std::min(4, 256 / Tc), passes = 4 / cpar, q = 8 * passes;
            if (passes > 1 && env_int("HGS_COL_XMAP", 1)) {
                double best = 0;
                int best_g = 0;
                for (int G = q; G <= cap && G / passes <= tiles; G += q) {
                    const int gp = G / passes, sweeps = (tiles + gp - 1) / gp;
                    const double eff = (double)tiles / ((double)sweeps * gp);
                    if (eff >= best - 1e-9) { best = eff; best_g = G; }
                }
                if (best_g > 0 && best >= 0.9) { col_blocks = best_g; col_xmap = 1; }
            }
        }
        // (8192 rows: one workgroup fits a CU, so 2 x #CU workgroups run as two rounds whose first tile each comes without
        //  the LDS staging of the previous one; one round of #CU workgroups with twice the tiles: cfg5pad 213.2 -> 209.6 us)
        tile_blocks = std::max(1, std::min(tiles, env_int("HGS_TILE_BLOCKS", g.Ph >= 8192 ? n_cu : n_cu * 2) / B));
        ew_blocks = (int)std::min<size_t>((P + 255) / 256, (size_t)std::max(1, n_cu * 8 / B));

        if (dalloc(&phase, B * S)) return HGS_ERR_DEVICE;
        lap("phase");
        if (dalloc(&gh, (size_t)B * g.Sh * g.Pw)) return HGS_ERR_DEVICE;
        lap("gh");
        if (dalloc(&w, B * P)) return HGS_ERR_DEVICE;
        if (dalloc(&t, B * P)) return HGS_ERR_DEVICE;
        lap("weights + target");
        if (dalloc(&wpartial, (size_t)B * std::max(std::max(col_blocks, tile_blocks), n_cu * 3))) return HGS_ERR_DEVICE;
        if (dalloc(&fpartial, (size_t)B * std::max(col_blocks, n_cu * 3))) return HGS_ERR_DEVICE;
        if (dalloc(&epartial, (size_t)B * ew_blocks)) return HGS_ERR_DEVICE;
        if (dalloc(&sums, (size_t)4 * B)) return HGS_ERR_DEVICE;
        if (dalloc(&wscale, (size_t)B)) return HGS_ERR_DEVICE;
        if (int e = fill_wscale_one()) return e;
        lap("partials");
        if (int e = make_twiddles(&tw_row, g.Pw)) return e;
        if (g.Ph == g.Pw) tw_col = tw_row;                    // square pads: one table
        else if (int e = make_twiddles(&tw_col, g.Ph)) return e;
        lap("twiddles");
        if (c.n_spots > 0) {
            if (dalloc(&spot_xy, (size_t)2 * c.n_spots)) return HGS_ERR_DEVICE;
            if (dalloc(&spot_amp, (size_t)c.n_spots)) return HGS_ERR_DEVICE;
            if (dalloc(&ext_amp, (size_t)c.n_spots)) return HGS_ERR_DEVICE;
            if (dalloc(&spot_fb, (size_t)B * c.n_spots)) return HGS_ERR_DEVICE;
        }
        lap("spots");
        amp_scalar = 1.0 / std::sqrt((double)S);  // Hologram.__init__ :401-402
        amp_norm2 = 1.0;
        HIPCHK(hipStreamSynchronize(stream));
        lap("sync");
        return 0;
    }

    // ---- general padded shapes (Bluestein) ----------------------------------------------------------
    // the longest line one workgroup transforms: 1024 lanes x 16 elements, LDS image (17/16) * 16384 * sizeof(C)
    static constexpr int max_line() { return sizeof(R) == 4 ? 16384 : 8192; }
    static bool axis_plain(int N) { return is_pow2(N) && N >= 256 && N <= max_line(); }
    static bool axis_ok(int N) { return N >= 2 && (axis_plain(N) || 2 * N - 1 <= max_line()); }
    static void host_fft(std::vector<std::complex<double>>& a) {      // in-place radix-2, forward sign
        const size_t n = a.size();
        for (size_t i = 1, j = 0; i < n; ++i) {
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            const double ang = -2.0 * M_PI / (double)len;
            for (size_t i = 0; i < n; i += len)
                for (size_t k = 0; k < len / 2; ++k) {
                    const std::complex<double> w(std::cos(ang * (double)k), std::sin(ang * (double)k));
                    const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                    a[i + k] = u + v;
                    a[i + k + len / 2] = u - v;
                }
        }
    }
    // tables of one centred transform of length N (bluestein.hpp): dir = -1 forward, +1 inverse
    // plain: no chirp (A = pre, Cc = post / sqrt(N), Bf unused but allocated as one element)
    int make_blue_tables(int N, int M, int dir, bool plain, C** out3) {
        using cd = std::complex<double>;
        const long long h = N / 2;
        const double sg = dir < 0 ? -1.0 : 1.0;
        auto unit = [&](long long num, long long den, double sign) {     // exp(sign * 2 pi i * num / den), num reduced first
            long long r = num % den;
            if (r < 0) r += den;
            const double a = sign * 2.0 * M_PI * (double)r / (double)den;
            return cd(std::cos(a), std::sin(a));
        };
        std::vector<cd> chirp(N), A(N), Cc(N), bt(plain ? 1 : M, cd(0, 0));
        for (long long n = 0; n < N; ++n) chirp[n] = plain ? cd(1, 0) : unit(n * n, 2LL * N, sg);   // W^(n^2/2), W = exp(sg 2 pi i / N)
        const double sc = 1.0 / std::sqrt((double)N);
        for (long long n = 0; n < N; ++n) {
            // forward: pre = W^(-n h), post = W^(h (k - h));  inverse (W -> conj W): pre = Wc^(n h), post = Wc^(-h (i + h))
            const cd pre = dir < 0 ? unit(-n * h, N, -1.0) : unit(n * h, N, 1.0);
            const cd post = dir < 0 ? unit(h * (n - h), N, -1.0) : unit(-h * (n + h), N, 1.0);
            A[n] = pre * chirp[n];
            Cc[n] = post * chirp[n] * sc;
        }
        if (!plain) {
            bt[0] = std::conj(chirp[0]);
            for (int d = 1; d < N; ++d) bt[d] = bt[M - d] = std::conj(chirp[d]);
            host_fft(bt);
            for (auto& v : bt) v /= (double)M;
        }
        auto up = [&](const std::vector<cd>& src, C** dst) -> int {
            std::vector<C> hbuf(src.size());
            for (size_t i = 0; i < src.size(); ++i) { hbuf[i].x = (R)src[i].real(); hbuf[i].y = (R)src[i].imag(); }
            HIPCHK(hipMalloc(reinterpret_cast<void**>(dst), hbuf.size() * sizeof(C)));
            return h2d(*dst, hbuf.data(), hbuf.size() * sizeof(C));
        };
        if (int e = up(A, &out3[0])) return e;
        if (int e = up(bt, &out3[1])) return e;
        return up(Cc, &out3[2]);
    }
    int init_general(const hgs_config& c) {
        auto conv_len = [](int N) { if (axis_plain(N)) return N; int M = 256; while (M < 2 * N - 1) M <<= 1; return M; };
        blue_M[0] = conv_len(g.Pw);
        blue_M[1] = conv_len(g.Ph);
        blue_plain[0] = axis_plain(g.Pw);
        blue_plain[1] = axis_plain(g.Ph);
        {   // one workgroup holds a whole line in LDS ((17/16) M complex numbers): fail here, with a message, rather
            // than at the first transform with a raw HIP error
            int lds_max = 0;
            HIPCHK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c.device));
            const size_t need = (size_t)(std::max(blue_M[0], blue_M[1]) / 16 * 17) * sizeof(C);
            if (need > (size_t)lds_max)
                return fail(HGS_ERR_UNSUPPORTED, "padded shape (%d, %d) needs %zu bytes of LDS per workgroup, the device offers %d",
                            c.pad_h, c.pad_w, need, lds_max);
        }
        ew_blocks = (int)std::min<size_t>((P + 255) / 256, (size_t)std::max(1, n_cu * 8 / B));
        col_blocks = tile_blocks = row_blocks = 1;
        if (dalloc(&phase, B * S)) return HGS_ERR_DEVICE;
        if (dalloc(&gh, (size_t)B * g.Sh * g.Pw)) return HGS_ERR_DEVICE;
        if (dalloc(&nfbuf, B * S)) return HGS_ERR_DEVICE;
        if (dalloc(&w, B * P)) return HGS_ERR_DEVICE;
        if (dalloc(&t, B * P)) return HGS_ERR_DEVICE;
        if (dalloc(&wpartial, (size_t)B * n_cu * 3)) return HGS_ERR_DEVICE;
        if (dalloc(&fpartial, (size_t)B * std::max(ew_blocks, n_cu * 3))) return HGS_ERR_DEVICE;
        if (dalloc(&epartial, (size_t)B * ew_blocks)) return HGS_ERR_DEVICE;
        if (dalloc(&sums, (size_t)4 * B)) return HGS_ERR_DEVICE;
        if (dalloc(&wscale, (size_t)B)) return HGS_ERR_DEVICE;
        if (int e = fill_wscale_one()) return e;
        for (int d = 0; d < 2; ++d) {
            const int N = d == 0 ? g.Pw : g.Ph;
            if (d == 1 && blue_M[1] == blue_M[0]) blue_tw[1] = blue_tw[0];
            else if (int e = make_twiddles(&blue_tw[d], blue_M[d])) return e;
            if (int e = make_blue_tables(N, blue_M[d], -1, blue_plain[d], blue_tab[d][0])) return e;
            if (int e = make_blue_tables(N, blue_M[d], +1, blue_plain[d], blue_tab[d][1])) return e;
        }
        if (c.n_spots > 0) {
            if (dalloc(&spot_xy, (size_t)2 * c.n_spots)) return HGS_ERR_DEVICE;
            if (dalloc(&spot_amp, (size_t)c.n_spots)) return HGS_ERR_DEVICE;
            if (dalloc(&ext_amp, (size_t)c.n_spots)) return HGS_ERR_DEVICE;
            if (dalloc(&spot_fb, (size_t)B * c.n_spots)) return HGS_ERR_DEVICE;
        }
        amp_scalar = 1.0 / std::sqrt((double)S);
        amp_norm2 = 1.0;
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    int blue_pass(int dim, int dir, const C* in, size_t in_line, size_t in_batch, int in_stride, int in_start, int in_len,
                  C* out, size_t out_line, size_t out_batch, int out_stride, int out_start, int out_len, int lines) {
        BlueArgs<R> a{};
        a.in = in; a.out = out; a.in_line = in_line; a.out_line = out_line; a.in_batch = in_batch; a.out_batch = out_batch;
        a.in_stride = in_stride; a.out_stride = out_stride; a.in_start = in_start; a.in_len = in_len;
        a.out_start = out_start; a.out_len = out_len; a.N = dim == 0 ? g.Pw : g.Ph;
        C** t3 = blue_tab[dim][dir < 0 ? 0 : 1];
        a.A = t3[0]; a.Bf = t3[1]; a.Cc = t3[2]; a.tw = blue_tw[dim];
        a.plain = blue_plain[dim] ? (dir < 0 ? 1 : 2) : 0;
        LCHK(launch_bluestein<R>(blue_M[dim], dim3(lines, B), stream, a));
        return 0;
    }
    int n2f_general(int store_pff) {
        if (int e = need_ff()) return e;
        if (store_pff) { if (int e = need_pff()) return e; }
        int r = timed(HGS_K_ROW, [&]() -> int {
            hipLaunchKernelGGL(gen_build_nearfield<R>, dim3((unsigned)((S + 255) / 256), B), dim3(256), 0, stream, (const R*)phase,
                               has_amp ? (const R*)amp : (const R*)nullptr, has_kern ? (const R*)kern : (const R*)nullptr,
                               (R)amp_scalar, S, nfbuf);
            HIPCHK(hipGetLastError());
            // rows: nearfield block (zero outside the SLM columns) -> G[r][kx]
            return blue_pass(0, -1, nfbuf, g.Sw, S, 1, g.c0, g.Sw, gh, g.Pw, (size_t)g.Sh * g.Pw, 1, 0, g.Pw, g.Sh);
        });
        if (r) return r;
        r = timed(HGS_K_COL_FWD, [&]() -> int {
            // columns: G[:, kx] (zero outside the SLM rows) -> farfield, column-major
            if (int e = blue_pass(1, -1, gh, 1, (size_t)g.Sh * g.Pw, g.Pw, g.r0, g.Sh, ff, g.Ph, P, 1, 0, g.Ph, g.Pw)) return e;
            hipLaunchKernelGGL(gen_amp_store<R>, dim3(ew_blocks, B), dim3(256), 0, stream, (const C*)ff, aff,
                               store_pff ? pff : (R*)nullptr, P, fpartial);
            HIPCHK(hipGetLastError());
            return 0;
        });
        if (r) return r;
        if (int e = reduce(fpartial, ew_blocks, sums + 0 * B)) return e;
        if (store_pff) have_pff = true;
        farfield_valid = true;
        return 0;
    }
    int f2n_general(bool complex_only) {
        if (!ff || !farfield_valid) return fail(HGS_ERR_STATE, "no farfield to transform back");
        int r = timed(HGS_K_COL_INV, [&]() -> int {
            return blue_pass(1, +1, ff, g.Ph, P, 1, 0, g.Ph, gh, 1, (size_t)g.Sh * g.Pw, g.Pw, g.r0, g.Sh, g.Pw);
        });
        if (r) return r;
        farfield_valid = false;
        return timed(HGS_K_ROW, [&]() -> int {
            if (int e = blue_pass(0, +1, gh, g.Pw, (size_t)g.Sh * g.Pw, 1, 0, g.Pw, nfbuf, g.Sw, S, 1, g.c0, g.Sw, g.Sh)) return e;
            if (!complex_only) {
                hipLaunchKernelGGL(gen_extract_phase<R>, dim3((unsigned)((S + 255) / 256), B), dim3(256), 0, stream, (const C*)nfbuf,
                                   has_kern ? (const R*)kern : (const R*)nullptr, S, phase);
                HIPCHK(hipGetLastError());
            }
            return 0;
        });
    }

    int init_compressed(const hgs_config& c) {
        if (c.n_spots < 1 || c.n_monomials < 1 || c.slm_h < 1 || c.slm_w < 1 || c.batch < 1)
            return fail(HGS_ERR_ARG, "compressed engine needs n_spots, n_monomials, slm shape and batch >= 1");
        opt_mono_tab = env_int("HGS_MONO_TAB", 1);
        g.Sh = c.slm_h; g.Sw = c.slm_w; g.r0 = g.c0 = 0;
        g.Ph = c.n_spots; g.Pw = 1;      // farfield-sized arrays are N-vectors; the transposes degenerate
        g.batch = B = c.batch;
        S = (size_t)g.Sh * g.Sw;
        P = (size_t)c.n_spots;
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        c_nblocks = (int)((S + (size_t)C_WG * C_PT - 1) / ((size_t)C_WG * C_PT));
        col_blocks = 1; tile_blocks = 1; row_blocks = 1;
        ew_blocks = (int)std::min<size_t>((P + 255) / 256, (size_t)std::max(1, n_cu * 8 / B));
        c_rows = std::max(6, c.n_monomials);
        if (dalloc(&phase, B * S)) return HGS_ERR_DEVICE;
        if (dalloc(&w, B * P)) return HGS_ERR_DEVICE;
        if (dalloc(&t, B * P)) return HGS_ERR_DEVICE;
        if (dalloc(&xg, S)) return HGS_ERR_DEVICE;
        if (dalloc(&yg, S)) return HGS_ERR_DEVICE;
        if (dalloc(&mono, (size_t)2 * c.n_monomials)) return HGS_ERR_DEVICE;
        if (dalloc(&coeff, (size_t)c_rows * P)) return HGS_ERR_DEVICE;
        run_rpr = (g.Sw + CR_RUN - 1) / CR_RUN;
        run_blocks = (g.Sh * run_rpr + 63) / 64;
        if (dalloc(&cpartial, (size_t)B * std::max(c_nblocks, run_blocks) * P)) return HGS_ERR_DEVICE;
        if (dalloc(&cnorm, (size_t)B * ((P + C_RED_SPOTS - 1) / C_RED_SPOTS))) return HGS_ERR_DEVICE;
        if (dalloc(&ext_amp, P)) return HGS_ERR_DEVICE;
        if (dalloc(&ext_r, B * P)) return HGS_ERR_DEVICE;
        if (dalloc(&epartial, (size_t)B * ew_blocks)) return HGS_ERR_DEVICE;
        if (dalloc(&wpartial, (size_t)B)) return HGS_ERR_DEVICE;
        if (dalloc(&fpartial, (size_t)B)) return HGS_ERR_DEVICE;
        if (dalloc(&sums, (size_t)4 * B)) return HGS_ERR_DEVICE;
        if (dalloc(&wscale, (size_t)B)) return HGS_ERR_DEVICE;
        if (int e = fill_wscale_one()) return e;
        amp_scalar = 1.0 / std::sqrt((double)S);
        amp_norm2 = 1.0;
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }

    // repack the monomial weights for the degree-specialised kernels (canonical [1,x,y,x2,xy,y2])
    int pack_coeff() {
        if (!has_mono || !has_coeff) return 0;
        const int M = cfg.n_monomials, N = cfg.n_spots;
        c_degree = 0;
        for (int m = 0; m < M; ++m) c_degree = std::max(c_degree, mono_host[2 * m] + mono_host[2 * m + 1]);
        for (int m = 0; m < M; ++m) {
            const int px = mono_host[2 * m], py = mono_host[2 * m + 1];
            if (px == -1 && py == 0) { c_degree = std::max(c_degree, 3); continue; }   // vortex plate: general kernels
            if (px < 0 || py < 0)
                return fail(HGS_ERR_ARG, "unrecognized term (%d, %d): the only pseudo-term is the vortex plate (-1, 0)", px, py);
        }
        if (c_degree <= 2) {
            std::vector<R> c6((size_t)6 * N, (R)0);
            for (int m = 0; m < M; ++m) {
                const int px = mono_host[2 * m], py = mono_host[2 * m + 1];
                const int slot = (px == 0 && py == 0) ? 0 : (px == 1 && py == 0) ? 1 : (px == 0 && py == 1) ? 2
                                 : (px == 2) ? 3 : (px == 1) ? 4 : 5;
                for (int n = 0; n < N; ++n) c6[(size_t)slot * N + n] += coeff_host[(size_t)m * N + n];
            }
            if (int e_ = h2d(coeff, c6.data(), c6.size() * sizeof(R))) return e_;
        } else {
            if (int e_ = h2d(coeff, coeff_host.data(), (size_t)M * N * sizeof(R))) return e_;
        }
        return sep_refresh();
    }

    // ---- separable / matrix-core form (compressed_sep.hpp) ----
    // split-K factor: fill the 2 * n_cu resident workgroups of cgemm_kouter in whole rounds (a 128 x 128
    // output grid rarely does by itself: 711 tiles = 1.39 rounds, 135 tiles = 0.26), smallest such split
    static void split_for(int tiles, int K, int n_cu, int* split, int* k_per) {
        const int slots = 2 * n_cu;
        int best = 1;
        double best_eff = 0;
        for (int sp = 1; sp <= 32 && K / sp >= 128; ++sp) {
            const int w = tiles * sp;
            const double eff = (double)w / ((double)((w + slots - 1) / slots) * slots);
            if (eff > best_eff + 0.08) { best_eff = eff; best = sp; }   // partial tiles cost traffic: need a real gain
        }
        int kp = ((K + best - 1) / best + CG_BK - 1) / CG_BK * CG_BK;
        *split = (K + kp - 1) / kp;
        *k_per = kp;
    }
    // called whenever the grids, the term set or the coefficients changed
    int sep_refresh() {
        if (int e = sep_refresh_impl()) return e;
        return run_refresh();
    }
    // Run kernels: x must be a regular grid over the columns (checked against the uploaded values), y a function of the
    // row, the polynomial of degree <= 2.  Per spot: canonical coefficients in double and the constant factor of the
    // recurrence (degree 2: C_n = exp(i 2 c3 h^2); degree 1: D_n = exp(i c1 h)).
    int run_refresh() {
        run_ok = false;
        if constexpr (sizeof(R) != 4) return 0;
        if (cfg.kind != 1) return 0;
        if (!(has_grid[0] && has_grid[1] && grid_sep[0] && grid_sep[1] && has_mono && has_coeff)) return 0;
        if (c_degree < 0 || c_degree > 2) return 0;
        const int M = cfg.n_monomials, N = cfg.n_spots, H = g.Sh, W = g.Sw;
        if (W < 2) return 0;
        const double x0 = xs_host[0], hx = (xs_host[W - 1] - x0) / (double)(W - 1);
        double xmax = 0;
        for (int x = 0; x < W; ++x) xmax = std::max(xmax, std::fabs(xs_host[x]));
        // the uploaded values are fp32 roundings of the grid: half an ulp of the largest coordinate each
        for (int x = 0; x < W; ++x)
            if (std::fabs(xs_host[x] - (x0 + x * hx)) > 2.5e-7 * xmax + 1e-30) return 0;
        std::vector<CRunRec> rec((size_t)N);
        for (int n = 0; n < N; ++n) {
            CRunRec& r = rec[n];
            for (double& c : r.c) c = 0;
            for (int m = 0; m < M; ++m) {
                const int px = mono_host[2 * m], py = mono_host[2 * m + 1];
                const int slot = (px == 0 && py == 0) ? 0 : (px == 1 && py == 0) ? 1 : (px == 0 && py == 1) ? 2
                                 : (px == 2) ? 3 : (px == 1) ? 4 : 5;
                r.c[slot] += (double)coeff_host[(size_t)m * N + n];
            }
            const double ang = c_degree == 2 ? 2.0 * r.c[3] * hx * hx : r.c[1] * hx;
            r.cr = (float)std::cos(ang);
            r.ci = (float)std::sin(ang);
            r.pad0 = r.pad1 = 0;
        }
        if (!run_rec) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&run_rec), (size_t)N * sizeof(CRunRec)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&run_ys), (size_t)H * sizeof(double)));
        }
        if (int e_ = h2d(run_rec, rec.data(), rec.size() * sizeof(CRunRec))) return e_;
        if (int e_ = h2d(run_ys, ys_host.data(), (size_t)H * sizeof(double))) return e_;
        run_x0 = x0;
        run_hx = hx;
        // spot chunks (grid.z): about eight waves per SIMD, whole groups of 64 spots
        int want = std::max(1, (8 * 4 * n_cu + run_blocks - 1) / run_blocks);
        want = std::min(want, 8);
        run_nper = std::max(64, ((N + want - 1) / want + 63) / 64 * 64);
        run_chunks = (N + run_nper - 1) / run_nper;
        run_ok = true;
        return 0;
    }
    bool use_run() const { return run_ok && opt_run; }
    unsigned bflag() const { return B > 1 ? DF_BATCH : 0u; }
    CRunArgs run_args(const CArgs<R>& a) {
        CRunArgs ra{};
        if constexpr (sizeof(R) == 4) ra.a = a;
        ra.a.nblocks = run_blocks;
        ra.rec = run_rec; ra.ys = run_ys; ra.x0 = run_x0; ra.hx = run_hx; ra.H = g.Sh; ra.W = g.Sw; ra.rpr = run_rpr;
        ra.n_per = run_nper; ra.nf_part = run_nf;
        return ra;
    }
    int run_n2f(CArgs<R>& a) {
        CRunArgs ra = run_args(a);
        a.nblocks = run_blocks;                       // what c_n2f_reduce sums over
        const dim3 grid(run_blocks, B, run_chunks);
        if (c_degree <= 1) { dispatch_note(dispatch_site<KCn2fRun, float, 1>(), bflag()); hipLaunchKernelGGL((c_n2f_run<1>), grid, dim3(64), 0, stream, ra); }
        else { dispatch_note(dispatch_site<KCn2fRun, float, 2>(), bflag()); hipLaunchKernelGGL((c_n2f_run<2>), grid, dim3(64), 0, stream, ra); }
        HIPCHK(hipGetLastError());
        return 0;
    }
    int run_f2n(const CArgs<R>& a) {
        if (!run_nf || run_nf_chunks < run_chunks) {
            if (run_nf) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(run_nf)); run_nf = nullptr; }
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&run_nf), (size_t)B * run_chunks * S * sizeof(Cx<float>)));
            run_nf_chunks = run_chunks;
        }
        CRunArgs ra = run_args(a);
        const dim3 grid(run_blocks, B, run_chunks);
        if (c_degree <= 1) { dispatch_note(dispatch_site<KCf2nRun, float, 1>(), bflag()); hipLaunchKernelGGL((c_f2n_run<1>), grid, dim3(64), 0, stream, ra); }
        else { dispatch_note(dispatch_site<KCf2nRun, float, 2>(), bflag()); hipLaunchKernelGGL((c_f2n_run<2>), grid, dim3(64), 0, stream, ra); }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(c_f2n_run_finish, dim3((unsigned)((S + 255) / 256), B), dim3(256), 0, stream, ra.a,
                           (const Cx<float>*)run_nf, run_chunks);
        HIPCHK(hipGetLastError());
        return 0;
    }
    int sep_refresh_impl() {
        c_sep = false;
        if (sizeof(R) != 4 || cfg.kind != 1) return 0;
        if (!(has_grid[0] && has_grid[1] && grid_sep[0] && grid_sep[1] && has_mono && has_coeff)) return 0;
        const int M = cfg.n_monomials, N = cfg.n_spots, H = g.Sh, W = g.Sw;
        int dx = 0, dy = 0;
        for (int m = 0; m < M; ++m) {
            const int px = mono_host[2 * m], py = mono_host[2 * m + 1];
            if (px < 0 || py < 0 || (px > 0 && py > 0) || px > SEP_MAXDEG || py > SEP_MAXDEG) return 0;   // not separable
            dx = std::max(dx, px);
            dy = std::max(dy, py);
        }
        std::vector<double> c((size_t)2 * (SEP_MAXDEG + 1) * N, 0.0);
        for (int m = 0; m < M; ++m) {
            const int px = mono_host[2 * m], py = mono_host[2 * m + 1];
            double* dst = (py == 0) ? &c[(size_t)px * N] : &c[((size_t)(SEP_MAXDEG + 1) + py) * N];
            for (int n = 0; n < N; ++n) dst[n] += (double)coeff_host[(size_t)m * N + n];
        }
        if (!sep_c) {
            // operands of the matrix-core GEMM are padded to whole tiles (zero filled once, never rewritten)
            auto up = [](int v, int q) { return (v + q - 1) / q * q; };
            // stream-K: the (tile, k tile) steps of each GEMM in 2 * #CU equal shares; per tile the first workgroup and the
            // number of partial planes it is spread over (consumers add exactly those)
            sk_tm1 = (N + CG_BM - 1) / CG_BM; sk_tn1 = (H + CG_BN - 1) / CG_BN; sk_kt1 = (W + CG_BK - 1) / CG_BK;
            sk_tm2 = (H + CG_BM - 1) / CG_BM; sk_tn2 = (W + CG_BN - 1) / CG_BN; sk_kt2 = (N + CG_BK - 1) / CG_BK;
            {
                const int t1 = sk_tm1 * sk_tn1, t2 = sk_tm2 * sk_tn2;
                std::vector<int> tab((size_t)2 * (t1 + t2));
                // (at most one workgroup per step: every workgroup owns work, the owners of a tile are consecutive)
                sk_G1 = (int)std::min<long long>(2 * n_cu, (long long)t1 * sk_kt1);
                sk_G2 = (int)std::min<long long>(2 * n_cu, (long long)t2 * sk_kt2);
                auto fill = [&](int tiles, int KT, int G, int* first, int* nseg) {
                    const long long total = (long long)tiles * KT;
                    int planes = 1;
                    for (int t = 0; t < tiles; ++t) {
                        first[t] = sk_owner((long long)t * KT, total, G);
                        nseg[t] = sk_owner((long long)(t + 1) * KT - 1, total, G) - first[t] + 1;
                        planes = std::max(planes, nseg[t]);
                    }
                    return planes;
                };
                sep_split1 = fill(t1, sk_kt1, sk_G1, tab.data(), tab.data() + t1);
                sep_split2 = fill(t2, sk_kt2, sk_G2, tab.data() + 2 * t1, tab.data() + 2 * t1 + t2);
                HIPCHK(hipMalloc(reinterpret_cast<void**>(&sk_tab), tab.size() * sizeof(int)));
                if (int e_ = h2d(sk_tab, tab.data(), tab.size() * sizeof(int))) return e_;
            }
            sep_Np = up(N, CG_BM); sep_Hp = up(H, CG_BN); sep_Wp = up(W, CG_BN);
            sep_Wk = sk_kt1 * CG_BK;
            sep_Nk = sk_kt2 * CG_BK;
            auto zalloc = [&](float2** p, size_t n) -> int {
                HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float2)));
                HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(float2), stream));
                return 0;
            };
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&sep_c), c.size() * sizeof(double)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&sep_g), (size_t)(W + H) * sizeof(double)));
            if (zalloc(&sep_ex, (size_t)sep_Nk * sep_Wp)) return HGS_ERR_DEVICE;       // [Nk][Wp]  (B of the f2n GEMM)
            if (zalloc(&sep_exT, (size_t)sep_Wk * sep_Np)) return HGS_ERR_DEVICE;      // [Wk][Np]  (A of the n2f GEMM)
            if (zalloc(&sep_ey, (size_t)N * H)) return HGS_ERR_DEVICE;
            if (zalloc(&sep_nfT, (size_t)B * sep_Wk * sep_Hp)) return HGS_ERR_DEVICE;  // [B][Wk][Hp]
            if (zalloc(&sep_b2, (size_t)B * sep_Nk * sep_Hp)) return HGS_ERR_DEVICE;   // [B][Nk][Hp]
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&sep_c1), (size_t)B * sk_tn1 * 2 * sep_split1 * sep_Np * sizeof(float2)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&sep_c2), (size_t)B * sep_split2 * H * W * sizeof(float2)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&sep_norm), (size_t)B * ((N + 255) / 256) * sizeof(double)));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(cgemm_streamk<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(cgemm_streamk<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES));
        }
        HIPCHK(hipMemcpyAsync(sep_c, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(sep_g, xs_host.data(), (size_t)W * sizeof(double), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(sep_g + W, ys_host.data(), (size_t)H * sizeof(double), hipMemcpyHostToDevice, stream));
        sep_degx = dx;
        sep_degy = dy;
        hipLaunchKernelGGL(sep_build_table, dim3((W + 255) / 256, N), dim3(256), 0, stream, (const double*)sep_c, dx, N,
                           (const double*)sep_g, W, sep_ex, sep_Wp, (size_t)sep_Nk * sep_Wp, sep_exT, sep_Np, (size_t)sep_Wk * sep_Np);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(sep_build_table, dim3((H + 255) / 256, N), dim3(256), 0, stream,
                           (const double*)(sep_c + (size_t)(SEP_MAXDEG + 1) * N), dy, N, (const double*)(sep_g + W), H, sep_ey,
                           H, (size_t)0, (float2*)nullptr, 0, (size_t)0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));   // c / xs_host are host temporaries
        c_sep = true;
        return 0;
    }
    bool use_sep() const {
        return c_sep && opt_separable && cfg.n_spots >= opt_sep_min;
    }
    // operands planar: real array at A / Bm, imaginary one planeA / planeB floats on.  E != nullptr: the epilogue contracts
    // the result with E over its columns into `part` instead of storing it (n2f)
    int launch_cgemm(const float* A, size_t planeA, const float* Bm, size_t planeB, float2* C, int M, int N, int KT, int lda, int ldb,
                     int tiles_m, int tiles_n, int planes, const int* first_wg, int G, size_t strideA, size_t strideB,
                     const float2* E = nullptr, int ldE = 0, float2* part = nullptr, int ldP = 0) {
        CgemmSkArgs a{A, A + planeA, Bm, Bm + planeB, C, M, N, KT, lda, ldb, tiles_m, tiles_n, planes, first_wg, strideA, strideB,
                      E, ldE, part, ldP};
        if (E) { dispatch_note(dispatch_site<KCgemm, float, 1>(), bflag()); hipLaunchKernelGGL(cgemm_streamk<1>, dim3(G, B), dim3(256), CG_LDS_BYTES, stream, a); }
        else { dispatch_note(dispatch_site<KCgemm, float, 0>(), bflag()); hipLaunchKernelGGL(cgemm_streamk<0>, dim3(G, B), dim3(256), CG_LDS_BYTES, stream, a); }
        HIPCHK(hipGetLastError());
        return 0;
    }
    // n2f: T = Ex^T-table x nf^T on the matrix cores, contracted with Ey over y in the GEMM's epilogue
    int sep_n2f() {
        const int N = cfg.n_spots, H = g.Sh, W = g.Sw;
        hipLaunchKernelGGL(sep_build_nft<R>, dim3((W + 31) / 32, (H + 31) / 32, B), dim3(32, 8), 0, stream, (const R*)phase,
                           has_amp ? (const R*)amp : (const R*)nullptr, has_kern ? (const R*)kern : (const R*)nullptr,
                           (R)amp_scalar, H, W, reinterpret_cast<float*>(sep_nfT), sep_Hp, (size_t)sep_Wk * sep_Hp);
        HIPCHK(hipGetLastError());
        const int t1 = sk_tm1 * sk_tn1;
        if (int e = launch_cgemm(reinterpret_cast<const float*>(sep_exT), (size_t)sep_Wk * sep_Np,
                                 reinterpret_cast<const float*>(sep_nfT), (size_t)sep_Wk * sep_Hp, nullptr, N, H, sk_kt1, sep_Np, sep_Hp,
                                 sk_tm1, sk_tn1, sep_split1, sk_tab, sk_G1, 0, (size_t)2 * sep_Wk * sep_Hp,
                                 (const float2*)sep_ey, H, sep_c1, sep_Np)) return e;
        const int nred = (N + 255) / 256;
        hipLaunchKernelGGL(sep_n2f_sum<R>, dim3(nred, B), dim3(256), 0, stream, (const float2*)sep_c1, sep_Np, sep_split1,
                           (const int*)(sk_tab + t1), sk_tm1, sk_tn1, N, 1.0 / std::sqrt((double)S), ff, sep_norm);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(c_n2f_finish<R>, dim3(B), dim3(1024), 0, stream, cargs(), (const double*)sep_norm, nred);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // f2n: conj(nf) = (conj(ff) Ey)^T x Ex on the matrix cores, then phase extraction (or the complex field)
    int sep_f2n(Cx<R>* nf_out) {
        const int N = cfg.n_spots, H = g.Sh, W = g.Sw;
        hipLaunchKernelGGL(sep_build_b2<R>, dim3((H + 255) / 256, N, B), dim3(256), 0, stream, (const Cx<R>*)ff,
                           (const float2*)sep_ey, N, H, reinterpret_cast<float*>(sep_b2), sep_Hp, (size_t)sep_Nk * sep_Hp);
        HIPCHK(hipGetLastError());
        const int t1 = sk_tm1 * sk_tn1, t2 = sk_tm2 * sk_tn2;
        if (int e = launch_cgemm(reinterpret_cast<const float*>(sep_b2), (size_t)sep_Nk * sep_Hp,
                                 reinterpret_cast<const float*>(sep_ex), (size_t)sep_Nk * sep_Wp, sep_c2, H, W, sk_kt2, sep_Hp, sep_Wp,
                                 sk_tm2, sk_tn2, sep_split2, sk_tab + 2 * t1, sk_G2, (size_t)2 * sep_Nk * sep_Hp, 0)) return e;
        hipLaunchKernelGGL(sep_f2n_finish<R>, dim3((unsigned)((S + 255) / 256), B), dim3(256), 0, stream, (const float2*)sep_c2,
                           sep_split2, (const int*)(sk_tab + 2 * t1 + t2), sk_tm2, W, S,
                           has_kern ? (const R*)kern : (const R*)nullptr, phase, nf_out);
        HIPCHK(hipGetLastError());
        return 0;
    }

    CArgs<R> cargs() {
        CArgs<R> a{};
        a.S = (int)S; a.N = cfg.n_spots; a.M = cfg.n_monomials; a.batch = B; a.xg = xg; a.yg = yg; a.mono = mono;
        a.coeff = coeff; a.phase = phase; a.amp = has_amp ? amp : nullptr; a.kern = has_kern ? kern : nullptr;
        a.amp_scalar = (R)amp_scalar; a.ff = ff; a.amp_ff = aff; a.partial = cpartial; a.nblocks = c_nblocks;
        a.fsum = sums + 0 * B; a.degree = c_degree;
        return a;
    }
    int compressed_ready() {
        if (!has_grid[0] || !has_grid[1] || !has_mono || !has_coeff)
            return fail(HGS_ERR_STATE, "compressed engine needs HGS_XGRID, HGS_YGRID, HGS_MONOMIALS and HGS_SPOT_COEFF");
        return 0;
    }
    int n2f_compressed(int store_pff) {
        if (int e = compressed_ready()) return e;
        if (int e = need_ff()) return e;
        if (store_pff) { if (int e = need_pff()) return e; }
        const int nred = (int)((P + C_RED_SPOTS - 1) / C_RED_SPOTS);
        int r = timed(HGS_K_COL_FWD, [&]() -> int {
            if (use_sep()) return sep_n2f();
            CArgs<R> a = cargs();
            const dim3 grid(c_nblocks, B);
            if (use_run()) { if (int e = run_n2f(a)) return e; }
            else if (c_degree <= 1) { dispatch_note(dispatch_site<KCn2fPix, R, 1>(), bflag()); hipLaunchKernelGGL((c_n2f_partial<R, 1>), grid, dim3(C_WG), 0, stream, a); }
            else if (c_degree == 2) { dispatch_note(dispatch_site<KCn2fPix, R, 2>(), bflag()); hipLaunchKernelGGL((c_n2f_partial<R, 2>), grid, dim3(C_WG), 0, stream, a); }
            else if (cfg.n_monomials <= C_MTAB && opt_mono_tab) { dispatch_note(dispatch_site<KCn2fPix, R, 3>(), bflag()); hipLaunchKernelGGL((c_n2f_partial<R, 3>), grid, dim3(C_WG), 0, stream, a); }
            else { dispatch_note(dispatch_site<KCn2fPix, R, 0>(), bflag()); hipLaunchKernelGGL((c_n2f_partial<R, 0>), grid, dim3(C_WG), 0, stream, a); }
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(c_n2f_reduce<R>, dim3(nred, B), dim3(C_RED_SPOTS * C_RED_SLICES), 0, stream, a, cnorm);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(c_n2f_finish<R>, dim3(B), dim3(1024), 0, stream, a, (const double*)cnorm, nred);
            HIPCHK(hipGetLastError());
            return 0;
        });
        if (r) return r;
        if (store_pff) {
            EwArgs<R> a{};
            a.P = P; a.batch = B; a.ff = ff; a.pff = pff;
            hipLaunchKernelGGL(ew_store_phase<R>, dim3(ew_blocks, B), dim3(256), 0, stream, a);
            HIPCHK(hipGetLastError());
            have_pff = true;
        }
        farfield_valid = true;
        return 0;
    }
    int f2n_compressed() {
        if (!ff || !farfield_valid) return fail(HGS_ERR_STATE, "no farfield to transform back");
        int r = timed(HGS_K_COL_INV, [&]() -> int {
            if (use_sep()) return sep_f2n(nullptr);
            CArgs<R> a = cargs();
            const dim3 grid(c_nblocks, B);
            if (use_run()) return run_f2n(a);
            if (c_degree <= 1) { dispatch_note(dispatch_site<KCf2nPix, R, 1>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 1>), grid, dim3(C_WG), 0, stream, a); }
            else if (c_degree == 2) { dispatch_note(dispatch_site<KCf2nPix, R, 2>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 2>), grid, dim3(C_WG), 0, stream, a); }
            else if (cfg.n_monomials <= C_MTAB && opt_mono_tab) { dispatch_note(dispatch_site<KCf2nPix, R, 3>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 3>), grid, dim3(C_WG), 0, stream, a); }
            else { dispatch_note(dispatch_site<KCf2nPix, R, 0>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 0>), grid, dim3(C_WG), 0, stream, a); }
            HIPCHK(hipGetLastError());
            return 0;
        });
        farfield_valid = false;
        return r;
    }

    // per-column fused launch: rule-specialised kernel where the pass is plain (fp32, no statistics, no extras)
    int fused_launch(int phase_mode, dim3 grid, const ColArgs<R>& a) {
        if constexpr (sizeof(R) == 4) {
            const bool plain = !a.do_stats && !a.cp.mraf && !a.cp.nog_pass && !a.cp.weights_only && a.cp.nog == nullptr;
            if (plain && opt_tile_rule) {
                if (!a.cp.do_update) return launch_fused_rule2(g.Ph, phase_mode, grid, stream, a);
                if (a.cp.method == HGS_WGS_LEONARDO || a.cp.method == HGS_WGS_KIM) return launch_fused_rule1(g.Ph, phase_mode, grid, stream, a);
            }
        }
        return launch_fused<R>(g.Ph, phase_mode, grid, stream, a);
    }
    // half-width tile-resident kernel (col_tile2_kernel): the grid it runs on, or 0 where it does not apply -- fp32, a dense
    // launch of a plain pass (Leonardo / Kim update or none; no statistics, MRAF, Nogrette sum or forward-only pass), and
    //   4096 rows: a batch (>= 2 holograms), farfield phase neither stored nor read (PHASE 0: the phase-storing instances do
    //              not fit 168 registers), SLM rows within six slots -- 3 x #CU workgroups over the batch, a multiple of 16
    //              per hologram so that the two halves of a tile run on one XCD together;
    //   2048 rows: SLM rows within ten slots -- 2 x #CU workgroups of two lane groups, one tile each at a time.
    int tile2_grid(bool sp, bool tile_path, const ColArgs<R>& a, int phase_mode) const {
        // (HGS_OPT_TILE_KERNEL = 0 forces the per-column kernel at every size: the tests' A/B reference)
        if (sizeof(R) != 4 || !opt_tile2 || !opt_tile || sp || a.do_stats || !opt_tile_rule) return 0;
        if (a.cp.mraf || a.cp.nog_pass || a.cp.weights_only || a.cp.nog != nullptr) return 0;
        if (a.cp.do_update && a.cp.method != HGS_WGS_LEONARDO && a.cp.method != HGS_WGS_KIM) return 0;
        const int nr = tile_slots();
        // the developer override HGS_TILE2_BLOCKS never exceeds what wpartial / the statistics partials are sized for
        // (B * max(col_blocks, tile_blocks, 3 * #CU) entries) nor the number of half tiles there are
        const int want = env_tile2_blocks > 0 ? std::min(env_tile2_blocks, 3 * n_cu) : 0;
        if (g.Ph == 4096) {
            if (B < opt_tile2_min_batch || (phase_mode != 0 && !(phase_mode == 2 && opt_tile2_phase2 && B == 1)) || !tile_path || !tile2_has(4096, nr)) return 0;
            const int per = (want > 0 ? want : 3 * n_cu) / B;
            return std::min(std::max(16, per / 16 * 16), std::max(16, g.Pw / 2 / 16 * 16));
        }
        if (g.Ph == 2048) {
            if (!tile2_has(2048, nr)) return 0;
            return std::max(1, std::min(g.Pw / 4, (want > 0 ? want : 2 * n_cu) / B));
        }
        return 0;
    }
    static int tile2_launch(int N, int phase, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int shift, int xmap) {
        return launch_tile2(N, phase, rule, nr, grid, s, a, shift, xmap);
    }
    static int tile2_launch(int, int, int, int, dim3, hipStream_t, const ColArgs<double>&, int, int) { return (int)hipErrorInvalidValue; }
    // (the tile-resident kernel is fp32 only; this branch is never taken for double)
    static int tile_rule(int N, int phase, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
        return a.col_list != nullptr ? launch_tile_rule_listed(N, phase, rule, nr, grid, s, a, m0) : launch_tile_rule(N, phase, rule, nr, grid, s, a, m0);
    }
    static int tile_rule(int, int, int, int, dim3, hipStream_t, const ColArgs<double>&, int) { return (int)hipErrorInvalidValue; }
    static int tile_split(int N, int phase, int nr, int rule_ok, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
        return a.do_stats ? launch_tile_split_stats(N, phase, nr, rule_ok, grid, s, a, m0) : launch_tile_split(N, phase, nr, rule_ok, grid, s, a, m0);
    }
    static int tile_split(int, int, int, int, dim3, hipStream_t, const ColArgs<double>&, int) { return (int)hipErrorInvalidValue; }
    static int presum_launch(int N, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) { return launch_presum(N, nr, grid, s, a, m0); }
    static int presum_launch(int, int, dim3, hipStream_t, const ColArgs<double>&, int) { return (int)hipErrorInvalidValue; }
    static int tile_presum(int N, int phase, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) { return launch_tile_presum(N, phase, nr, grid, s, a, m0); }
    static int tile_presum(int, int, int, dim3, hipStream_t, const ColArgs<double>&, int) { return (int)hipErrorInvalidValue; }
    static int row_split_launch(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<float>& a) { return launch_row_split(N, mode, grid, s, a); }
    static int row_split_launch(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<double>& a) { return launch_row_split(N, mode, grid, s, a); }

    int fill_wscale_one() {
        hipLaunchKernelGGL(set_scalar<R>, dim3((B + 63) / 64), dim3(64), 0, stream, wscale, B, (R)1);
        HIPCHK(hipGetLastError());
        w_pending = false;
        w_unit = false;
        return 0;
    }

    // ---- lazily allocated farfield-sized buffers ----
    int need_ff() {
        if (!ff) { if (dalloc(&ff, B * P)) return HGS_ERR_DEVICE; }
        if (!aff) { if (dalloc(&aff, B * P)) return HGS_ERR_DEVICE; }
        return 0;
    }
    int need_aff() {
        if (!aff) { if (dalloc(&aff, B * P)) return HGS_ERR_DEVICE; }
        return 0;
    }
    int need_pff() {
        if (!pff) { if (dalloc(&pff, B * P)) return HGS_ERR_DEVICE; }
        return 0;
    }
    int need_staging() {
        if (!staging) HIPCHK(hipMalloc(&staging, B * P * sizeof(C)));
        return 0;
    }
    int need_zw() {
        if (!zw) { if (dalloc(&zw, B * P)) return HGS_ERR_DEVICE; }
        return 0;
    }

    // ---- profiling wrapper ----
    template <typename F> int timed(int kind, F&& f) {
        if (!prof) return f();
        Ev e;
        e.kind = kind;
        HIPCHK(hipEventCreate(&e.a));
        HIPCHK(hipEventCreate(&e.b));
        HIPCHK(hipEventRecord(e.a, stream));
        int r = f();
        HIPCHK(hipEventRecord(e.b, stream));
        evs.push_back(e);
        return r;
    }
    int profile_enable(int on) override { prof = on != 0; return 0; }
    int profile_read(double* out) override {
        HIPCHK(hipStreamSynchronize(stream));
        for (auto& e : evs) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            prof_ms[e.kind] += ms;
            prof_n[e.kind] += 1;
            hipEventDestroy(e.a);
            hipEventDestroy(e.b);
        }
        evs.clear();
        for (int k = 0; k < HGS_K_COUNT; ++k) {
            out[2 * k] = prof_ms[k];
            out[2 * k + 1] = prof_n[k];
            prof_ms[k] = prof_n[k] = 0;
        }
        return 0;
    }

    // ---- natural <-> column-major moves through the staging buffer ----
    template <typename E> int upload_T(E* dst, const void* host, size_t nbytes, bool src_device = false) {
        // host natural [B][Ph][Pw] -> device column-major [B][Pw][Ph]
        const size_t one = P * sizeof(E);
        if (nbytes != one * B && nbytes != one) return fail(HGS_ERR_ARG, "array size %zu does not match %zu x {1,%d}", nbytes, one, B);
        if (int e = need_staging()) return e;
        E* st = reinterpret_cast<E*>(staging);
        for (int b = 0; b < B; ++b) {
            const char* src = (const char*)host + (nbytes == one ? 0 : (size_t)b * one);
            HIPCHK(hipMemcpyAsync(st + (size_t)b * P, src, one, src_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        }
        dim3 grid((g.Pw + 31) / 32, (g.Ph + 31) / 32, B);
        hipLaunchKernelGGL((transpose_scale<E, R>), grid, dim3(32, 8), 0, stream, (const E*)st, dst, g.Ph, g.Pw,
                           (const R*)nullptr, cfg.kind == 0 ? g.lane_T : 0, 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    template <typename E> int download_T(const E* src, void* dst, size_t nbytes, bool dst_device, const R* scale) {
        const size_t all = P * sizeof(E) * B;
        if (nbytes != all) return fail(HGS_ERR_ARG, "array size %zu does not match %zu", nbytes, all);
        if (int e = need_staging()) return e;
        E* st = reinterpret_cast<E*>(staging);
        dim3 grid((g.Ph + 31) / 32, (g.Pw + 31) / 32, B);
        hipLaunchKernelGGL((transpose_scale<E, R>), grid, dim3(32, 8), 0, stream, src, st, g.Pw, g.Ph, scale,
                           cfg.kind == 0 ? g.lane_T : 0, 0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dst, st, all, dst_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }

    int set_array(int which, const void* host, size_t nbytes, bool src_device) override {
        if (which == HGS_PHASE || which == HGS_AMP || which == HGS_AMP_SCALAR || which == HGS_PROP_KERNEL) gh_state = -1;
        if (which == HGS_PROP_KERNEL && nbytes == 0) {       // "no kernel" (Hologram.propagation_kernel = None / 0)
            has_kern = false;
            farfield_valid = false;
            return 0;
        }
        if (!host) return fail(HGS_ERR_ARG, "null source pointer");
        const hipMemcpyKind kind = src_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (src_device && (which == HGS_XGRID || which == HGS_YGRID || which == HGS_MONOMIALS || which == HGS_SPOT_COEFF ||
                           which == HGS_SPOT_INDEX || which == HGS_AMP_SCALAR))
            return fail(HGS_ERR_UNSUPPORTED, "array selector %d is parsed on the host: upload it with hgs_set_array", which);
        switch (which) {
            case HGS_PHASE: {
                const size_t one = S * sizeof(R);
                if (nbytes != one * B && nbytes != one) return fail(HGS_ERR_ARG, "phase: bad size %zu", nbytes);
                for (int b = 0; b < B; ++b)
                    HIPCHK(hipMemcpyAsync(phase + (size_t)b * S, (const char*)host + (nbytes == one ? 0 : b * one), one, kind, stream));
                HIPCHK(hipStreamSynchronize(stream));
                farfield_valid = false;
                return 0;
            }
            case HGS_AMP: {
                if (nbytes != S * sizeof(R)) return fail(HGS_ERR_ARG, "amp: bad size %zu", nbytes);
                if (!amp) HIPCHK(hipMalloc(reinterpret_cast<void**>(&amp), S * sizeof(R)));
                if (int e_ = h2d(amp, host, nbytes, src_device)) return e_;
                has_amp = true;
                double s = 0;
                if (src_device) {         // ||amp||^2 on the device: per-block partials (double), folded here
                    const int nb = (int)std::min<size_t>((S + 255) / 256, 1024);
                    double* part = nullptr;
                    HIPCHK(hipMalloc(reinterpret_cast<void**>(&part), (size_t)nb * sizeof(double)));
                    hipLaunchKernelGGL(ew_sumsq<R>, dim3(nb, 1), dim3(256), 0, stream, (const R*)amp, S, part);
                    if (hipGetLastError() != hipSuccess) { hipFree(part); return fail(HGS_ERR_DEVICE, "amplitude norm launch failed"); }
                    std::vector<double> hp(nb);
                    hipError_t e1 = hipMemcpyAsync(hp.data(), part, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, stream);
                    hipError_t e2 = hipStreamSynchronize(stream);
                    hipFree(part);
                    if (e1 != hipSuccess || e2 != hipSuccess) return fail(HGS_ERR_DEVICE, "amplitude norm failed");
                    for (double v : hp) s += v;
                } else {
                    const R* h = (const R*)host;
                    // (NaN entries skipped, like ew_sumsq on the device path and the reference's nansum normalisation)
                    for (size_t i = 0; i < S; ++i) { const double v = (double)h[i]; if (v == v) s += v * v; }
                }
                amp_norm2 = s;
                farfield_valid = false;
                return 0;
            }
            case HGS_AMP_SCALAR: {
                if (nbytes != sizeof(R)) return fail(HGS_ERR_ARG, "amp scalar: bad size %zu", nbytes);
                amp_scalar = (double)*(const R*)host;
                has_amp = false;
                amp_norm2 = amp_scalar * amp_scalar * (double)S;
                farfield_valid = false;
                return 0;
            }
            case HGS_PROP_KERNEL: {
                if (nbytes != S * sizeof(R)) return fail(HGS_ERR_ARG, "propagation kernel: bad size %zu", nbytes);
                if (!kern) HIPCHK(hipMalloc(reinterpret_cast<void**>(&kern), S * sizeof(R)));
                if (int e_ = h2d(kern, host, nbytes, src_device)) return e_;
                has_kern = true;
                farfield_valid = false;
                return 0;
            }
            case HGS_TARGET:
                has_target = true;
                sparse_dirty = true;
                return upload_T<R>(t, host, nbytes, src_device);
            case HGS_WEIGHTS: {
                sparse_dirty = true;
                int e = upload_T<R>(w, host, nbytes, src_device);
                if (e) return e;
                return fill_wscale_one();
            }
            case HGS_PHASE_FF: {
                if (int e = need_pff()) return e;
                have_pff = true;
                return upload_T<R>(pff, host, nbytes, src_device);
            }
            case HGS_ZERO_WEIGHTS: {
                if (int e = need_zw()) return e;
                return upload_T<C>(zw, host, nbytes, src_device);
            }
            case HGS_XGRID:
            case HGS_YGRID: {
                if (cfg.kind != 1) return fail(HGS_ERR_STATE, "grids belong to the compressed engine");
                if (nbytes != S * sizeof(R)) return fail(HGS_ERR_ARG, "grid: bad size %zu", nbytes);
                if (int e_ = h2d(which == HGS_XGRID ? xg : yg, host, nbytes)) return e_;
                has_grid[which == HGS_XGRID ? 0 : 1] = true;
                farfield_valid = false;
                {   // product grid?  x depends on the column only, y on the row only
                    const R* h = (const R*)host;
                    const int H = g.Sh, W = g.Sw;
                    bool sep = true;
                    if (which == HGS_XGRID) {
                        for (int y = 1; y < H && sep; ++y)
                            for (int x = 0; x < W; ++x)
                                if (h[(size_t)y * W + x] != h[x]) { sep = false; break; }
                        xs_host.assign(W, 0.0);
                        for (int x = 0; x < W; ++x) xs_host[x] = (double)h[x];
                    } else {
                        for (int y = 0; y < H && sep; ++y)
                            for (int x = 1; x < W; ++x)
                                if (h[(size_t)y * W + x] != h[(size_t)y * W]) { sep = false; break; }
                        ys_host.assign(H, 0.0);
                        for (int y = 0; y < H; ++y) ys_host[y] = (double)h[(size_t)y * W];
                    }
                    grid_sep[which == HGS_XGRID ? 0 : 1] = sep;
                }
                return sep_refresh();
            }
            case HGS_MONOMIALS: {
                if (cfg.kind != 1) return fail(HGS_ERR_STATE, "monomials belong to the compressed engine");
                if (nbytes != (size_t)2 * cfg.n_monomials * sizeof(int32_t)) return fail(HGS_ERR_ARG, "monomials: bad size");
                const int32_t* h = (const int32_t*)host;
                if (has_mono && std::equal(mono_host.begin(), mono_host.end(), h)) return 0;   // unchanged term set
                mono_host.assign(h, h + 2 * cfg.n_monomials);
                if (int e_ = h2d(mono, host, nbytes)) return e_;
                has_mono = true;
                farfield_valid = false;
                return pack_coeff();
            }
            case HGS_SPOT_COEFF: {
                if (cfg.kind != 1) return fail(HGS_ERR_STATE, "spot coefficients belong to the compressed engine");
                if (nbytes != (size_t)cfg.n_monomials * cfg.n_spots * sizeof(R)) return fail(HGS_ERR_ARG, "spot coefficients: bad size");
                const R* h = (const R*)host;
                coeff_host.assign(h, h + (size_t)cfg.n_monomials * cfg.n_spots);
                has_coeff = true;
                farfield_valid = false;
                return pack_coeff();
            }
            case HGS_SPOT_INDEX: {
                if (cfg.n_spots <= 0) return fail(HGS_ERR_STATE, "engine was created with n_spots = 0");
                if (nbytes != (size_t)2 * cfg.n_spots * sizeof(int32_t)) return fail(HGS_ERR_ARG, "spot index: bad size");
                const int32_t* h = (const int32_t*)host;
                const int hw = 0;  // bounds are checked against the window at constraint time
                for (int n = 0; n < cfg.n_spots; ++n)
                    if (h[n] < hw || h[n] >= g.Pw || h[cfg.n_spots + n] < 0 || h[cfg.n_spots + n] >= g.Ph)
                        return fail(HGS_ERR_ARG, "spot %d outside the computational grid", n);
                if (int e_ = h2d(spot_xy, host, nbytes)) return e_;
                spot_xy_host.assign(h, h + 2 * cfg.n_spots);
                has_spots = true;
                return 0;
            }
            case HGS_SPOT_AMP:
            case HGS_EXTERNAL_AMP: {
                if (cfg.kind == 1 && which == HGS_SPOT_AMP) return fail(HGS_ERR_ARG, "compressed targets are set with HGS_TARGET");
                if (cfg.n_spots <= 0) return fail(HGS_ERR_STATE, "engine was created with n_spots = 0");
                if (nbytes != (size_t)cfg.n_spots * sizeof(double)) return fail(HGS_ERR_ARG, "spot amplitudes: bad size");
                if (int e_ = h2d(which == HGS_SPOT_AMP ? spot_amp : ext_amp, host, nbytes, src_device)) return e_;
                return 0;
            }
        }
        return fail(HGS_ERR_ARG, "unknown array selector %d", which);
    }
    std::vector<int32_t> spot_xy_host;

    int phase_ref(PhaseRef* o) override {
        o->ptr = phase; o->S = S; o->B = B; o->real_bytes = (int)sizeof(R); o->device = cfg.device; o->stream = stream;
        return 0;
    }
    // phase <- the phase another engine holds, device to device (no host bounce); one source hologram broadcasts
    int copy_phase_from(const PhaseRef& src) override {
        if (src.S != S || src.real_bytes != (int)sizeof(R) || (src.B != B && src.B != 1))
            return fail(HGS_ERR_ARG, "hgs_copy_phase: engines differ in SLM shape, precision or batch");
        if (hipStreamSynchronize(src.stream) != hipSuccess) return fail(HGS_ERR_DEVICE, "hgs_copy_phase: source stream sync failed");
        for (int b = 0; b < B; ++b) {
            const char* from = (const char*)src.ptr + (src.B == 1 ? 0 : (size_t)b * S * sizeof(R));
            if (src.device == cfg.device) HIPCHK(hipMemcpyAsync(phase + (size_t)b * S, from, S * sizeof(R), hipMemcpyDeviceToDevice, stream));
            else HIPCHK(hipMemcpyPeerAsync(phase + (size_t)b * S, cfg.device, from, src.device, S * sizeof(R), stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        farfield_valid = false;
        gh_state = -1;
        return 0;
    }

    int normalize_weights_now() {
        // fold the pending 1/||w|| into the stored weights (general path keeps them normalised)
        if (!w_pending) return 0;
        hipLaunchKernelGGL(scale_weights_kernel<R>, dim3(ew_blocks, B), dim3(256), 0, stream, w, (const R*)wscale, P);
        HIPCHK(hipGetLastError());
        return fill_wscale_one();
    }
    int get_array(int which, void* dst, size_t nbytes, bool dst_device) override {
        if (!dst) return fail(HGS_ERR_ARG, "null destination pointer");
        switch (which) {
            case HGS_PHASE: {
                if (nbytes != S * sizeof(R) * B) return fail(HGS_ERR_ARG, "phase: bad size %zu", nbytes);
                HIPCHK(hipMemcpyAsync(dst, phase, nbytes, dst_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                return 0;
            }
            case HGS_PHASE_PREV: {
                if (!phase_prev || !have_prev) return fail(HGS_ERR_STATE, "no previous phase is held (HGS_OPT_KEEP_PREV_PHASE, one-iteration fused calls)");
                if (nbytes != S * sizeof(R) * B) return fail(HGS_ERR_ARG, "previous phase: bad size %zu", nbytes);
                HIPCHK(hipMemcpyAsync(dst, phase_prev, nbytes, dst_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                return 0;
            }
            case HGS_TARGET: return download_T<R>(t, dst, nbytes, dst_device, nullptr);
            case HGS_WEIGHTS: return download_T<R>(w, dst, nbytes, dst_device, w_pending ? wscale : nullptr);
            case HGS_PHASE_FF:
                if (!pff || !have_pff) return fail(HGS_ERR_STATE, "phase_ff has not been computed");
                return download_T<R>(pff, dst, nbytes, dst_device, nullptr);
            case HGS_FARFIELD:
                if (!ff || !farfield_valid) return fail(HGS_ERR_STATE, "farfield is not materialised (call hgs_nearfield2farfield)");
                return download_T<C>(ff, dst, nbytes, dst_device, nullptr);
            case HGS_AMP_FF:
                if (!aff || !farfield_valid) return fail(HGS_ERR_STATE, "amp_ff is not materialised (call hgs_nearfield2farfield)");
                return download_T<R>(aff, dst, nbytes, dst_device, nullptr);
            case HGS_ZERO_WEIGHTS:
                if (!zw) return fail(HGS_ERR_STATE, "zero_weights not allocated");
                return download_T<C>(zw, dst, nbytes, dst_device, nullptr);
        }
        return fail(HGS_ERR_ARG, "array selector %d cannot be read back", which);
    }

    int reset_weights() override {
        sparse_dirty = true;
        hipLaunchKernelGGL(reset_weights_kernel<R>, dim3(ew_blocks * B), dim3(256), 0, stream, w, (const R*)t, zw, B * P);
        HIPCHK(hipGetLastError());
        return fill_wscale_one();
    }
    // Hologram.reset (:442-478): weights from the target, phase_ff / farfield / amp_ff back to "None"
    int reset_state() override {
        have_pff = false;
        have_prev = false;
        farfield_valid = false;
        gh_state = -1;          // (a kept G is the un-extracted phasor of the last body: a reset hologram starts from its phase, like a new one)
        return reset_weights();
    }
    // n values at listed pixels, `0` everywhere else (SpotHologram targets: n_spots numbers instead of P)
    int set_array_sparse(int which, const int32_t* xy, const void* values, int n) override {
        if (cfg.kind != 0) return fail(HGS_ERR_UNSUPPORTED, "sparse uploads are for the padded-grid holograms");
        if (which != HGS_TARGET && which != HGS_WEIGHTS) return fail(HGS_ERR_ARG, "sparse upload: HGS_TARGET or HGS_WEIGHTS only");
        if (n < 0 || (n > 0 && (!xy || !values))) return fail(HGS_ERR_ARG, "sparse upload: null list");
        // duplicates: the last entry wins (NumPy fancy assignment); resolved here so that the scatter is order-free
        std::vector<uint32_t> pos;
        std::vector<R> val;
        pos.reserve(n); val.reserve(n);
        {
            std::vector<std::pair<uint32_t, int>> key(n);
            for (int k = 0; k < n; ++k) {
                const int x = xy[k], y = xy[n + k];
                if (x < 0 || x >= g.Pw || y < 0 || y >= g.Ph) return fail(HGS_ERR_ARG, "sparse upload: pixel %d outside the grid", k);
                const int yd = col_pos(y, g.lane_T);
                key[k] = {(uint32_t)((size_t)x * g.Ph + yd), k};
            }
            std::stable_sort(key.begin(), key.end(), [](const std::pair<uint32_t, int>& a, const std::pair<uint32_t, int>& b) { return a.first < b.first; });
            for (int k = 0; k < n; ++k)
                if (k + 1 == n || key[k + 1].first != key[k].first) {
                    pos.push_back(key[k].first);
                    val.push_back(static_cast<const R*>(values)[key[k].second]);
                }
        }
        R* dst = which == HGS_TARGET ? t : w;
        HIPCHK(hipMemsetAsync(dst, 0, (size_t)B * P * sizeof(R), stream));
        const int m = (int)pos.size();
        if (m > 0) {
            uint32_t* dpos = nullptr;
            R* dval = nullptr;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&dpos), (size_t)m * sizeof(uint32_t)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&dval), (size_t)m * sizeof(R)));
            hipError_t e1 = hipMemcpyAsync(dpos, pos.data(), (size_t)m * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
            hipError_t e2 = hipMemcpyAsync(dval, val.data(), (size_t)m * sizeof(R), hipMemcpyHostToDevice, stream);
            if (e1 == hipSuccess && e2 == hipSuccess) {
                hipLaunchKernelGGL(scatter_values<R>, dim3((m + 255) / 256, B), dim3(256), 0, stream, dst, (const uint32_t*)dpos,
                                   (const R*)dval, m, P);
                e1 = hipGetLastError();
            }
            hipStreamSynchronize(stream);
            hipFree(dpos);
            hipFree(dval);
            if (e1 != hipSuccess || e2 != hipSuccess) return fail(HGS_ERR_DEVICE, "sparse upload failed");
        } else {
            HIPCHK(hipStreamSynchronize(stream));
        }
        sparse_dirty = true;
        if (which == HGS_TARGET) { has_target = true; return 0; }
        return fill_wscale_one();
    }

    // ---- operator launches ----
    RowArgs<R> row_args(bool finalize) {
        RowArgs<R> a{};
        a.g = g; a.phase = phase; a.amp = has_amp ? amp : nullptr; a.kern = has_kern ? kern : nullptr;
        a.amp_scalar = (R)amp_scalar; a.gh = gh; a.tw = tw_row; a.scale = (R)(1.0 / std::sqrt((double)g.Pw));
        a.wpartial = finalize ? wpartial : nullptr; a.n_wpartial = wpartial_n; a.wscale = wscale;
        a.xcd_map = row_xcd;
        a.n_row_blocks = row_blocks;
        {   // shifted form of the row kernel: fp32, one-row workgroups, the SLM columns within eight register slots
            const int T = g.Pw / 16;
            const int s0 = g.c0 / T, s1 = (g.c0 + g.Sw - 1) / T;
            a.shifted = ((sizeof(R) == 4 || opt_row_shift64) && g.Pw >= 4096 && s1 - s0 + 1 <= 8 && opt_row_shift) ? 1 : 0;
            a.m0 = s0;
        }
        return a;
    }
    // load / store: 0 = every column, 1 = active columns, 2 = active columns dilated by the spot windows
    // mode: row_kernel MODE (3 = MODE 2 that also writes the phase; float32 only)
    int run_row(int mode, bool finalize, int load_sparse = 0, int store_sparse = 0) {
        gh_state = -1;
        int r_ = run_row_impl(mode, finalize, load_sparse, store_sparse);
        if (r_ == 0 && mode != 1) gh_state = store_sparse;      // gh holds G of the columns this launch stored
        return r_;
    }
    // (fused loops only; p = the plan of the call's first iteration)
    template <typename PlanT> int keep_prev_phase(const PlanT& p, int n) {
        if (!opt_prev_phase) return 0;
        if (n != 1) { have_prev = false; return 0; }      // the phases in between are never materialised
        if (p.use_fixed) return 0;                          // phase_ff is not rewritten: what is held stays what describes it
        if (!phase_prev) { if (dalloc(&phase_prev, (size_t)B * S)) return HGS_ERR_DEVICE; }
        HIPCHK(hipMemcpyAsync(phase_prev, phase, (size_t)B * S * sizeof(R), hipMemcpyDeviceToDevice, stream));
        have_prev = true;
        return 0;
    }
    bool gh_holds(int need) const { return opt_keep_g && (gh_state == need || gh_state == 0 || (gh_state == 2 && need == 1 && dil_valid)); }
    int run_row_impl(int mode, bool finalize, int load_sparse, int store_sparse) {
        return timed(HGS_K_ROW, [&]() -> int {
            RowArgs<R> a = row_args(finalize);
            a.load_mask = load_sparse == 1 ? lane_mask : load_sparse == 2 ? lane_mask_d : nullptr;
            a.store_mask = store_sparse == 1 ? lane_mask : store_sparse == 2 ? lane_mask_d : nullptr;
            // one extra (row-less) block folds the weight-norm partials when asked to
            // dense fp32 launches between iterations at 4096: workgroups walk several rows, next row prefetched into LDS
            int blocks = row_blocks;
            // (measured, tools/row_tail_probe.py: it pays where a one-row-per-workgroup launch ends in a partial round that the
            //  walk turns into a third row for a quarter to a half of the workgroups -- 1152 rows 28.1 -> 26.3 us, 1280 rows
            //  29.1 -> 27.5 us, 1040 / 1088 / 1200 / 1248 rows 1.0 - 1.7 us ahead; level at 1024 rows, behind at 1312 and 1536)
            if (sizeof(R) == 4 && g.Pw == 4096 && mode == 2 && opt_row_pref && row_blocks_pref > 0 && !a.load_mask && !a.store_mask &&
                (B == 1 ? (g.Sh > 2 * row_blocks_pref && 2 * g.Sh <= 5 * row_blocks_pref) : g.Sh >= 4 * row_blocks_pref)) {
                a.prefetch = 1;
                a.n_row_blocks = blocks = row_blocks_pref;
            }
            if (row_split) {          // single-pass MRAF: H = gh * wscale + gh2 (wscale final: scale_from_sum ran)
                a.prefetch = 0;
                a.n_row_blocks = blocks = row_blocks;
                a.gh2 = gh2;
                // (a column-list launch already reads the listed columns only, and a noise box fills its list: there the
                //  second mask costs its fetch -- 42.4 -> 46.1 us at cfg 5 -- and saves nothing)
                a.gh2_mask = (row_split_noise_only || (opt_gh2_mask && !sparse_dirty && !a.load_mask)) ? lane_mask_noise : nullptr;
                row_split = false;
                row_split_noise_only = false;
                LCHK(row_split_launch(g.Pw, mode, dim3(blocks, B), stream, a));
                return 0;
            }
            LCHK(launch_row<R>(g.Pw, mode, dim3(blocks + (finalize ? 1 : 0), B), stream, a));
            return 0;
        });
    }
    // active columns dilated by the offsets [lo, hi] of the spot integration window
    int refresh_dilated(int lo, int hi) {
        if (dil_valid && lo == dil_lo && hi == dil_hi) return 0;
        if (gh_state == 2) gh_state = -1;
        if (!col_active_d) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&col_active_d), (size_t)B * g.Pw));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&col_list_d), (size_t)B * g.Pw * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&n_active_d_dev), (size_t)B * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&lane_mask_d), (size_t)B * (g.Pw / 16) * sizeof(unsigned short)));
        }
        hipLaunchKernelGGL(dilate_active_cols, dim3((g.Pw + 255) / 256, B), dim3(256), 0, stream,
                           (const unsigned char*)col_active, g.Pw, lo, hi, col_active_d);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(compact_active_cols, dim3(B), dim3(256), 0, stream, (const unsigned char*)col_active_d, g.Pw,
                           col_list_d, n_active_d_dev, lane_mask_d);
        HIPCHK(hipGetLastError());
        std::vector<int> h(B);
        HIPCHK(hipMemcpyAsync(h.data(), n_active_d_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        n_active_d_max = *std::max_element(h.begin(), h.end());
        dil_lo = lo; dil_hi = hi; dil_valid = true;
        return 0;
    }
    // the tile-resident fused column kernel applies: fp32, 4096 / 8192 rows, the SLM rows within six register slots
    // (the kernel shifts its transform input by tile_shift() rows -- any multiple of 16 keeps the shift-theorem factor a
    //  per-lane constant -- so the SLM rows start in the first 16 rows of register slot 0)
    int tile_shift() const { return opt_tile_shift16 ? (g.r0 / 16) * 16 : (g.r0 / (g.Ph / 16)) * (g.Ph / 16); }
    int tile_slots() const { const int Tc = g.Ph / 16; return (g.r0 - tile_shift() + g.Sh + Tc - 1) / Tc; }
    bool tile_geometry_ok() const {
        if (sizeof(R) != 4 || g.Ph < 4096 || !opt_tile) return false;
        return tile_slots() <= 6;
    }
    // columns a workgroup pass of the column kernels handles side by side (ColCfg<N>::CPAR)
    int col_cpar() const {
        const int T = g.Ph / 16;
        return T >= 256 ? 1 : std::min(4, 256 / T);
    }
    int list_blocks(int n_list) const { return std::max(1, std::min((n_list + col_cpar() - 1) / col_cpar(), n_cu * 3)); }
    // (re)build the active-column list when weights or target changed since the last scan
    int refresh_sparse() {
        if (!sparse_dirty) return 0;
        if (gh_state > 0) gh_state = -1;       // (a G stored on the old column lists; one of every column stays good)
        if (!col_active) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&col_active), (size_t)B * g.Pw));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&sig_rows), (size_t)B * g.Pw * sizeof(unsigned short)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&col_list), (size_t)B * g.Pw * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&n_active_dev), (size_t)B * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&lane_mask), (size_t)B * (g.Pw / 16) * sizeof(unsigned short)));
        }
        hipLaunchKernelGGL(scan_active_cols<R>, dim3(g.Pw, B), dim3(256), 0, stream, (const R*)w, (const R*)t, g.Ph, g.Pw,
                           col_active, g.lane_T > 0 ? sig_rows : (unsigned short*)nullptr);
        HIPCHK(hipGetLastError());
        // Where the tile-resident kernel can run the column pass and the active columns fill their 4-column tiles at least
        // half (images, MRAF noise boxes -- not spot arrays, whose columns sit alone in their tiles), the active set is
        // rounded to whole tiles and that kernel walks the tile list: 45 ns per column against 80 for the per-column
        // kernel at 8192 points, and the row kernel moves whole 32-byte tile rows.
        sparse_tiles = false;
        if (tile_geometry_ok() && opt_tile_list) {
            std::vector<unsigned char> act((size_t)B * g.Pw);
            HIPCHK(hipMemcpyAsync(act.data(), col_active, act.size(), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            bool dense = true;
            for (int b = 0; b < B && dense; ++b) {
                int n_col = 0, n_tile = 0;
                for (int c = 0; c < g.Pw; c += 4) {
                    const unsigned char* q = act.data() + (size_t)b * g.Pw + c;
                    const int k = (q[0] != 0) + (q[1] != 0) + (q[2] != 0) + (q[3] != 0);
                    n_col += k;
                    n_tile += k > 0;
                }
                dense = n_col > 0 && n_col >= 2 * n_tile;
            }
            if (dense) {
                for (size_t c = 0; c < act.size(); c += 4) {
                    const unsigned char on = (act[c] | act[c + 1] | act[c + 2] | act[c + 3]) ? 1 : 0;
                    for (int k = 0; k < 4; ++k) act[c + k] |= on;          // (bits 1, 2 of scan_active_cols stay per column)
                }
                HIPCHK(hipMemcpyAsync(col_active, act.data(), act.size(), hipMemcpyHostToDevice, stream));
                HIPCHK(hipStreamSynchronize(stream));      // (act goes out of scope)
                sparse_tiles = true;
            }
        }
        hipLaunchKernelGGL(compact_active_cols, dim3(B), dim3(256), 0, stream, (const unsigned char*)col_active, g.Pw,
                           col_list, n_active_dev, lane_mask);
        HIPCHK(hipGetLastError());
        if (!lane_mask_noise) HIPCHK(hipMalloc(reinterpret_cast<void**>(&lane_mask_noise), (size_t)B * (g.Pw / 16) * sizeof(unsigned short)));
        hipLaunchKernelGGL(flag_lane_mask, dim3((g.Pw / 16 + 255) / 256, B), dim3(256), 0, stream, (const unsigned char*)col_active,
                           g.Pw, 4, lane_mask_noise);
        HIPCHK(hipGetLastError());
        std::vector<int> h(B);
        HIPCHK(hipMemcpyAsync(h.data(), n_active_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        n_active_max = *std::max_element(h.begin(), h.end());
        n_active_min = *std::min_element(h.begin(), h.end());
        sparse_dirty = false;
        dil_valid = false;
        noise_valid = false;
        signal_valid = false;
        return 0;
    }
    // the columns that hold a finite non-zero target as a list (the per-column pre-pass of the single-inverse MRAF update)
    int refresh_signal() {
        if (int e = refresh_sparse()) return e;
        if (signal_valid) return 0;
        if (!col_list_signal) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&col_list_signal), (size_t)B * g.Pw * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&n_signal_dev), (size_t)B * sizeof(int)));
        }
        if (!lane_mask_tmp) HIPCHK(hipMalloc(reinterpret_cast<void**>(&lane_mask_tmp), (size_t)B * (g.Pw / 16) * sizeof(unsigned short)));
        hipLaunchKernelGGL(compact_active_cols, dim3(B), dim3(256), 0, stream, (const unsigned char*)col_active, g.Pw,
                           col_list_signal, n_signal_dev, lane_mask_tmp, 2);
        HIPCHK(hipGetLastError());
        std::vector<int> h(B);
        HIPCHK(hipMemcpyAsync(h.data(), n_signal_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        n_signal_max = *std::max_element(h.begin(), h.end());
        signal_valid = true;
        return 0;
    }
    // per-column single-pass MRAF: the columns that hold a NaN target as a list (the buffer of the noise part is zeroed by the
    // caller once it decides for the single pass: its NaN-target pixels are rewritten by every pass, everything else must read
    // as zero, also after the target moved)
    int refresh_noise() {
        if (int e = refresh_sparse()) return e;
        if (noise_valid) return 0;
        if (!col_list_noise) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&col_list_noise), (size_t)B * g.Pw * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&n_noise_dev), (size_t)B * sizeof(int)));
        }
        if (!lane_mask_tmp) HIPCHK(hipMalloc(reinterpret_cast<void**>(&lane_mask_tmp), (size_t)B * (g.Pw / 16) * sizeof(unsigned short)));
        ffb_zeroed = false;
        hipLaunchKernelGGL(compact_active_cols, dim3(B), dim3(256), 0, stream, (const unsigned char*)col_active, g.Pw,
                           col_list_noise, n_noise_dev, lane_mask_tmp, 4);
        HIPCHK(hipGetLastError());
        std::vector<int> h(B);
        HIPCHK(hipMemcpyAsync(h.data(), n_noise_dev, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        n_noise_max = *std::max_element(h.begin(), h.end());
        noise_valid = true;
        return 0;
    }
    ColArgs<R> col_args() {
        ColArgs<R> a{};
        a.g = g; a.gh = gh; a.ff = ff; a.amp_ff = aff; a.pff = pff; a.w = w; a.t = t; a.wscale = wscale;
        a.wpartial = wpartial; a.fpartial = fpartial; a.tw = tw_col; a.scale = (R)(1.0 / std::sqrt((double)g.Ph));
        a.col_xmap = col_xmap;
        // float64, 4096 / 8192 rows: col_fused_kernel in its shifted form (the SLM rows in the first fnr register slots)
        if (sizeof(R) == 8 && g.Ph >= 4096 && opt_fused_shift) {
            const int sh = (g.r0 / 16) * 16, Tc = g.Ph / 16, nr = (g.r0 - sh + g.Sh + Tc - 1) / Tc;
            if (nr <= 6) { a.fshift = sh; a.fnr = nr; }
        }
        return a;
    }
    int reduce(const double* partial, int n, double* out) {
        hipLaunchKernelGGL(reduce_partials, dim3(B), dim3(256), 0, stream, partial, n, out);
        HIPCHK(hipGetLastError());
        return 0;
    }

    int n2f(int store_pff) override {
        RoctxRange range(opt_roctx, "hgs_nearfield2farfield");
        row_split = row_split_noise_only = false;       // (a single-pass MRAF call that failed between its column and row launch must not leak)
        if (cfg.kind == 1) return n2f_compressed(store_pff);
        if (general) return n2f_general(store_pff);
        if (int e = need_ff()) return e;
        if (store_pff) { if (int e = need_pff()) return e; }
        // (a fused loop that ended on its dense row launch left G of every column behind: the transform starts with its column pass)
        if (!(opt_keep_g && gh_state == 0)) { if (int e = run_row(0, false)) return e; }
        int r = timed(HGS_K_COL_FWD, [&]() -> int {
            ColArgs<R> a = col_args();
            a.store_pff = store_pff;
            LCHK(launch_col<R>(g.Ph, C_FWD | C_STORE, dim3(col_blocks, B), stream, a));
            return 0;
        });
        if (r) return r;
        if (int e = reduce(fpartial, col_blocks, sums + 0 * B)) return e;
        if (store_pff) have_pff = true;
        farfield_valid = true;
        return 0;
    }

    int f2n() override {
        RoctxRange range(opt_roctx, "hgs_farfield2nearfield");
        row_split = row_split_noise_only = false;
        if (cfg.kind == 1) return f2n_compressed();
        if (general) return f2n_general(false);
        if (!ff || !farfield_valid) return fail(HGS_ERR_STATE, "no farfield to transform back");
        int r = timed(HGS_K_COL_INV, [&]() -> int {
            gh_state = -1;
            LCHK(launch_col<R>(g.Ph, C_LOAD | C_INV, dim3(col_blocks, B), stream, col_args()));
            return 0;
        });
        if (r) return r;
        farfield_valid = false;  // farfield now holds the constrained field, phase moves on
        return run_row(1, false);
    }

    // Hologram._farfield2nearfield(extract=False) (:1058-1073): the complex nearfield over the SLM,
    // kept on the device for MultiplaneHologram's weighted sum
    int f2n_complex() override {
        if (!ff || !farfield_valid) return fail(HGS_ERR_STATE, "no farfield to transform back");
        if (!nfbuf) { if (dalloc(&nfbuf, B * S)) return HGS_ERR_DEVICE; }
        if (general) return f2n_general(true);
        if (cfg.kind == 1) {
            int r = timed(HGS_K_COL_INV, [&]() -> int {
                if (use_sep()) return sep_f2n(nfbuf);
                CArgs<R> a = cargs();
                a.nf_out = nfbuf;
                const dim3 grid(c_nblocks, B);
                if (use_run()) return run_f2n(a);
                if (c_degree <= 1) { dispatch_note(dispatch_site<KCf2nPix, R, 1>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 1>), grid, dim3(C_WG), 0, stream, a); }
                else if (c_degree == 2) { dispatch_note(dispatch_site<KCf2nPix, R, 2>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 2>), grid, dim3(C_WG), 0, stream, a); }
                else if (cfg.n_monomials <= C_MTAB && opt_mono_tab) { dispatch_note(dispatch_site<KCf2nPix, R, 3>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 3>), grid, dim3(C_WG), 0, stream, a); }
                else { dispatch_note(dispatch_site<KCf2nPix, R, 0>(), bflag()); hipLaunchKernelGGL((c_f2n<R, 0>), grid, dim3(C_WG), 0, stream, a); }
                HIPCHK(hipGetLastError());
                return 0;
            });
            farfield_valid = false;
            return r;
        }
        int r = timed(HGS_K_COL_INV, [&]() -> int {
            gh_state = -1;
            LCHK(launch_col<R>(g.Ph, C_LOAD | C_INV, dim3(col_blocks, B), stream, col_args()));
            return 0;
        });
        if (r) return r;
        farfield_valid = false;
        return timed(HGS_K_ROW, [&]() -> int {
            RowArgs<R> a = row_args(false);
            a.nf_out = nfbuf;
            LCHK(launch_row<R>(g.Pw, 1, dim3(row_blocks, B), stream, a));
            return 0;
        });
    }
    int mp_info(MpInfo* o) override {
        o->nf = nfbuf; o->kern = has_kern ? kern : nullptr; o->phase = phase; o->S = S; o->B = B;
        o->real_bytes = (int)sizeof(R); o->device = cfg.device; o->stream = stream;
        return 0;
    }
    int mp_combine(const MpInfo* infos, const double* weights, int n) override {
        MpArgs<R> a{};
        a.n = n; a.S = S; a.total = (size_t)B * S;
        for (int k = 0; k < n; ++k) {
            a.nf[k] = static_cast<const C*>(infos[k].nf);
            a.kern[k] = static_cast<const R*>(infos[k].kern);
            a.phase[k] = static_cast<R*>(infos[k].phase);
            a.w[k] = (R)weights[k];
        }
        hipLaunchKernelGGL(multiplane_combine<R>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, stream, a);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }

    // flag evolution of _gs_farfield_routines (:1552-1585); returns what this iteration must do
    struct Plan { int do_update, use_fixed, store_phase; };
    Plan plan_iteration(hgs_step* st, uint8_t* hist_slot) {
        Plan p{0, 0, 0};
        // _update_stats ran before the routines: the history sees the flag as it is now (:1479)
        if (hist_slot) *hist_slot = st->fixed_phase ? 1 : 0;
        if (st->false_run >= 0) st->false_run = st->fixed_phase ? 0 : st->false_run + 1;
        const bool wgs = st->method != HGS_GS;
        if (wgs && st->iter > 0) {
            p.do_update = 1;
            if (st->method == HGS_WGS_KIM) {
                const bool was_not_fixed = !st->fixed_phase;
                if (was_not_fixed && st->iter >= st->fix_phase_iteration - 1 &&
                    st->false_run >= st->fix_phase_iteration)
                    st->fixed_phase = 1;
                if ((st->fixed_phase && !have_pff) || was_not_fixed) p.store_phase = 1;
            } else {
                st->fixed_phase = 0;
            }
        }
        // :1601  "if not fixed or phase_ff is None: phase_ff = atan2(F)".  A phase stored in this very
        // iteration equals atan2(F), so "store + rebuild from F" is the same thing as using it.
        if (st->fixed_phase && !have_pff) p.store_phase = 1;
        p.use_fixed = (st->fixed_phase && have_pff && !p.store_phase) ? 1 : 0;
        return p;
    }

    // WGS-Kim fixed by efficiency (_hologram.py:1560-1569).  In the reference the flag is raised inside the
    // routines of the iteration whose recorded efficiency exceeds the threshold; that iteration still takes its
    // phase from the current farfield and stores it (was_not_fixed, :1582), so raising the flag right AFTER the
    // iteration is the same thing -- and the statistics the fused pass accumulates are available by then.
    static bool eff_gate_set(const hgs_step* st) {
        return st->method == HGS_WGS_KIM && st->fix_phase_efficiency == st->fix_phase_efficiency;
    }
    // the efficiency recorded for the iteration that just ran (st->iter not yet advanced); eff_host: value already on
    // the host (general path) or null to read slot `dev` on the device
    int eff_gate_after(hgs_step* st, const double* eff_host, const double* dev) {
        if (!eff_gate_set(st) || st->fixed_phase || st->iter <= 0) return 0;
        double eff = 0;
        if (eff_host) eff = *eff_host;
        else {
            HIPCHK(hipMemcpyAsync(&eff, dev, sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        if (eff > st->fix_phase_efficiency) st->fixed_phase = 1;
        return 0;
    }
    int eff_gate_check(const hgs_step* st, int groups) {
        if (!eff_gate_set(st)) return 0;
        if (st->efficiency_group < 0 || st->efficiency_group > 1 || !(groups & (1 << st->efficiency_group)))
            return fail(HGS_ERR_ARG, "Must track statistics to fix phase based on efficiency!");
        if (B != 1) return fail(HGS_ERR_UNSUPPORTED, "fix_phase_efficiency needs one flag per hologram: batch must be 1");
        return 0;
    }

    CParams<R> cparams(const hgs_step* st, const Plan& p) {
        CParams<R> c{};
        c.method = st->method; c.do_update = p.do_update; c.use_fixed = p.use_fixed; c.store_phase = p.store_phase;
        c.mraf = st->mraf_enabled; c.has_mraf_factor = st->has_mraf_factor; c.zero_mode = st->zero_mode;
        c.p_exp = (R)st->feedback_exponent; c.p_fac = (R)st->feedback_factor;
        c.mraf_factor = (R)st->mraf_factor; c.zero_factor = (R)st->zero_factor;
        c.inv_fnorm = (R)(1.0 / std::sqrt(amp_norm2));  // Parseval: ||F|| = ||nearfield|| = ||amp||
        c.log2_inv_fnorm = (R)(-0.5 * std::log2(amp_norm2));
        return c;
    }

    int check_step(const hgs_step* st) {
        if (st->method < HGS_GS || st->method > HGS_WGS_TANH) return fail(HGS_ERR_ARG, "unknown method %d", st->method);
        if (st->feedback < HGS_FB_PIXEL || st->feedback > HGS_FB_EXTERNAL) return fail(HGS_ERR_ARG, "unknown feedback %d", st->feedback);
        if (!has_target) return fail(HGS_ERR_STATE, "target has not been set");
        if (cfg.kind == 0 && st->feedback != HGS_FB_PIXEL && st->method != HGS_GS) {
            if (!has_spots) return fail(HGS_ERR_STATE, "spot feedback needs HGS_SPOT_INDEX / HGS_SPOT_AMP");
            if (st->feedback == HGS_FB_SPOT_WINDOW) {
                if (st->spot_window < 1) return fail(HGS_ERR_ARG, "spot_window must be >= 1");
                const int flo = (int)std::floor(-(st->spot_window - 1) / 2.0);
                const int fhi = flo + st->spot_window - 1;
                for (int n = 0; n < cfg.n_spots; ++n) {
                    const int x = spot_xy_host[n], y = spot_xy_host[cfg.n_spots + n];
                    if (x + flo < 0 || y + flo < 0 || x + fhi >= g.Pw || y + fhi >= g.Ph)
                        return fail(HGS_ERR_ARG, "integration window of spot %d leaves the grid (IndexError in the reference)", n);
                }
            }
        }
        return 0;
    }

    // general constraint on the materialised farfield
    int constraint_planned(hgs_step* st, const Plan& p) {
        if (!ff || !farfield_valid) return fail(HGS_ERR_STATE, "farfield is not materialised");
        if (int e = need_pff()) return e;
        if (int e = normalize_weights_now()) return e;
        if (p.do_update) sparse_dirty = true;     // the general rules rewrite the weight array
        if (st->mraf_enabled && st->zero_mode) { if (int e = need_zw()) return e; }
        if (st->mraf_enabled && st->fixed_phase && !have_pff && !p.store_phase)
            return fail(HGS_ERR_STATE, "fixed_phase with MRAF needs a stored phase_ff (reference quirk A12)");
        CParams<R> cp = cparams(st, p);
        EwArgs<R> a{};
        a.P = P; a.batch = B; a.ff = ff; a.amp_ff = aff; a.pff = pff; a.w = w; a.t = t;
        a.fsum = sums + 0 * B; a.nogsum = sums + 1 * B; a.partial = epartial; a.wsum = nullptr; a.zero_weights = zw;
        a.cp = cp;
        const dim3 eg(ew_blocks, B), eb(256);
        return timed(HGS_K_ELEMENTWISE, [&]() -> int {
            bool pixel_update = false;
            if (p.do_update && cfg.kind == 1 && st->feedback == HGS_FB_EXTERNAL) {
                // CompressedSpotHologram "external_spot" (_spots.py:978-989): the N-vector rule with
                // external_spot_amp as the feedback amplitude
                hipLaunchKernelGGL(convert_d2r<R>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream,
                                   (const double*)ext_amp, ext_r, (int)P, B);
                HIPCHK(hipGetLastError());
                hipLaunchKernelGGL(ew_sumsq<R>, eg, eb, 0, stream, (const R*)ext_r, P, epartial);
                HIPCHK(hipGetLastError());
                if (int e = reduce(epartial, ew_blocks, sums + 3 * B)) return e;
                a.amp_ff = ext_r;
                a.fsum = sums + 3 * B;
            }
            if (p.do_update) {
                if (st->feedback == HGS_FB_PIXEL || cfg.kind == 1) {
                    if (st->method == HGS_WGS_NOGRETTE) {
                        hipLaunchKernelGGL(ew_nogrette_sum<R>, eg, eb, 0, stream, a);
                        HIPCHK(hipGetLastError());
                        if (int e = reduce(epartial, ew_blocks, sums + 1 * B)) return e;
                    }
                    hipLaunchKernelGGL(ew_weight_update<R>, eg, eb, 0, stream, a);
                    HIPCHK(hipGetLastError());
                    if (int e = reduce(epartial, ew_blocks, sums + 2 * B)) return e;
                    pixel_update = true;
                } else {
                    SpotArgs<R> s{};
                    s.g = g; s.n_spots = cfg.n_spots; s.width = st->spot_window; s.feedback = st->feedback;
                    s.spot_xy = spot_xy; s.amp_ff = aff; s.ext_amp = ext_amp; s.spot_amp = spot_amp; s.w = w;
                    s.fb = spot_fb; s.cp = cp;
                    if (st->feedback == HGS_FB_SPOT_WINDOW) {
                        hipLaunchKernelGGL(spot_window<R>, dim3((cfg.n_spots + 127) / 128, B), dim3(128), 0, stream, s);
                        HIPCHK(hipGetLastError());
                    }
                    hipLaunchKernelGGL(spot_update<R>, dim3(B), dim3(256), 0, stream, s);
                    HIPCHK(hipGetLastError());
                }
            }
            a.wsum = pixel_update ? sums + 2 * B : nullptr;
            // the rebuild recomputes and stores phase_ff whenever it is not "use_fixed"
            hipLaunchKernelGGL(ew_rebuild<R>, eg, eb, 0, stream, a);
            HIPCHK(hipGetLastError());
            have_pff = true;
            return 0;
        });
    }

    int constraint(hgs_step* st) override {
        RoctxRange range(opt_roctx, "hgs_farfield_constraint");
        if (int e = check_step(st)) return e;
        if (eff_gate_set(st) && st->iter > 0)     // the stepwise caller evaluates the gate itself (it holds the statistics)
            return fail(HGS_ERR_ARG, "Must track statistics to fix phase based on efficiency!");
        Plan p = plan_iteration(st, nullptr);
        return constraint_planned(st, p);
    }

    bool fused_ok(const hgs_step* st) const {
        // MRAF rides the fused kernels unless the zero region carries zero_weights feedback (:1613-1616)
        return cfg.kind == 0 && !general && !(st->mraf_enabled && st->zero_mode) && st->feedback == HGS_FB_PIXEL;
    }

    // ---- spot feedback on sparse targets ("computational_spot" / "external_spot", _spots.py:1573-1624) ----
    // The N-vector weight rule needs |F|^2 only in the w x w windows around the spots, so the forward
    // transform runs on the spot columns dilated by the window (col_kernel FWD|STORE over a list), the
    // existing window-sum / N-vector kernels update the weights at the spot pixels, and the constrained
    // field is formed and transformed back by the fused kernel over the spot columns (weight update
    // off).  Everything else of the farfield is exactly zero and is neither computed nor moved.
    bool spot_sparse_ok(const hgs_step* st) {
        if (cfg.kind != 0 || general || st->mraf_enabled || st->method == HGS_GS || st->feedback == HGS_FB_PIXEL) return false;
        if (!opt_sparse || opt_stepwise) return false;
        if (refresh_sparse()) return false;
        return n_active_min > 0 && n_active_max * 4 <= g.Pw;
    }
    int iterate_spot_sparse(hgs_step* st, int n, uint8_t* hist) {
        if (int e = normalize_weights_now()) return e;      // the N-vector rule keeps the weights normalised
        if (int e = need_ff()) return e;
        const int groups = stat_ctx ? stat_ctx->groups : 0;
        auto window = [](int w, int* lo, int* hi) { *lo = (int)std::floor(-(w - 1) / 2.0); *hi = *lo + w - 1; };
        int lo = 0, hi = 0, l2, h2;
        if (st->feedback == HGS_FB_SPOT_WINDOW) window(st->spot_window, &lo, &hi);
        // the statistics windows sit at floor(spot_knm) (analysis.take, quirk A18), the weights at rint(spot_knm):
        // one more column to the left
        if (groups & 2) { window(stat_ctx->width, &l2, &h2); lo = std::min(lo, l2 - 1); hi = std::max(hi, h2); }
        if (int e = refresh_dilated(lo, hi)) return e;
        auto windows_needed = [&](const Plan& q) {
            return (st->feedback == HGS_FB_SPOT_WINDOW && q.do_update) || (groups & 2);
        };
        farfield_valid = false;
        w_unit = false;           // (spot_update writes the weights itself)
        Plan p = plan_iteration(st, hist ? hist : nullptr);
        if (int e = keep_prev_phase(p, n)) return e;
        if (!gh_holds(windows_needed(p) ? 2 : 1)) { if (int e = run_row(0, false, 0, windows_needed(p) ? 2 : 1)) return e; }
        for (int i = 0; i < n; ++i) {
            gh_state = -1;
            if (p.use_fixed || p.store_phase) { if (int e = need_pff()) return e; }
            const CParams<R> cp = cparams(st, p);
            if (windows_needed(p)) {
                int r = timed(HGS_K_COL_FWD, [&]() -> int {
                    ColArgs<R> a = col_args();
                    a.col_list = col_list_d;
                    a.n_active = n_active_d_dev;
                    LCHK(launch_col<R>(g.Ph, C_FWD | C_STORE, dim3(list_blocks(n_active_d_max), B), stream, a));
                    return 0;
                });
                if (r) return r;
            }
            if (p.do_update) {
                int r = timed(HGS_K_ELEMENTWISE, [&]() -> int {
                    SpotArgs<R> sa{};
                    sa.g = g; sa.n_spots = cfg.n_spots; sa.width = st->spot_window; sa.feedback = st->feedback;
                    sa.spot_xy = spot_xy; sa.amp_ff = aff; sa.ext_amp = ext_amp; sa.spot_amp = spot_amp; sa.w = w;
                    sa.fb = spot_fb; sa.cp = cp;
                    sa.inline_window = 1;
                    hipLaunchKernelGGL(spot_update<R>, dim3(B), dim3(256), 0, stream, sa);
                    HIPCHK(hipGetLastError());
                    return 0;
                });
                if (r) return r;
            }
            int r = timed(HGS_K_COL_FUSED, [&]() -> int {
                ColArgs<R> a = col_args();
                a.cp = cp;
                a.cp.do_update = 0;                          // the weights were updated above
                a.col_list = col_list;
                a.n_active = n_active_dev;
                const int blocks = list_blocks(n_active_max);
                wpartial_n = blocks;
                const int phase_mode = p.use_fixed ? 2 : (p.store_phase ? 1 : 0);
                if (groups & 1) {
                    a.do_stats = 1;
                    a.spartial = stat_partial;
                    a.tsum = stat_tsum;
                    a.inv_fsum = 1.0 / amp_norm2;
                    hipLaunchKernelGGL(stat_fill_neutral, dim3((unsigned)((stat_nslots + 255) / 256)), dim3(256), 0, stream,
                                       stat_partial, stat_nslots);
                    LCHK(launch_fused_stats<R>(g.Ph, phase_mode, dim3(blocks, B), stream, a));
                } else {
                    LCHK(fused_launch(phase_mode, dim3(blocks, B), a));
                }
                return 0;
            });
            if (r) return r;
            if (stat_ctx) { if (int e = fused_stats_finish(i)) return e; }
            if (p.store_phase) have_pff = true;
            if (stat_ctx) { if (int e = eff_gate_after(st, nullptr, stat_ctx->dev_out + ((size_t)i * 2 + st->efficiency_group) * B * 4)) return e; }
            st->iter++;
            Plan pn{0, 0, 0};
            int store_next = 1;
            if (i + 1 < n) {
                pn = plan_iteration(st, hist ? hist + i + 1 : nullptr);
                store_next = windows_needed(pn) ? 2 : 1;
            }
            const int last_mode = (opt_keep_g && sizeof(R) == 4) ? 3 : 1;          // (MODE 3 stores every column, see iterate())
            if (int e = run_row(i + 1 < n ? 2 : last_mode, false, 1, i + 1 < n ? store_next : (last_mode == 3 ? 0 : store_next))) return e;
            p = pn;
        }
        return 0;
    }

    int iterate(hgs_step* st, int n, uint8_t* hist) override {
        RoctxRange range(opt_roctx, "hgs_iterate");
        row_split = row_split_noise_only = false;
        if (n < 0) return fail(HGS_ERR_ARG, "n_iter must be >= 0");
        if (n == 0) return 0;
        if (int e = check_step(st)) return e;
        // the reference raises as soon as an iteration with iter > 0 consults statistics that were never tracked
        if (!stat_ctx && eff_gate_set(st) && (st->iter > 0 || n > 1))
            return fail(HGS_ERR_ARG, "Must track statistics to fix phase based on efficiency!");
        const bool fused = fused_ok(st) && !opt_stepwise;
        if (!fused && spot_sparse_ok(st)) return iterate_spot_sparse(st, n, hist);
        if (!fused) {
            have_prev = false;           // the general operators keep HGS_PHASE_FF itself up to date
            for (int i = 0; i < n; ++i) {
                if (int e = n2f(0)) return e;
                Plan p = plan_iteration(st, hist ? hist + i : nullptr);
                if (int e = constraint_planned(st, p)) return e;
                if (int e = f2n()) return e;
                st->iter++;
            }
            return 0;
        }
        farfield_valid = false;
        // Sparse targets: when few columns hold a non-zero weight/target, only those columns are
        // transformed (col_fused_kernel with a column list) and only they cross HBM between the two
        // kernels.  phase_ff (WGS-Kim) is then stored on the active columns only: nothing else can be
        // read back by the loop.
        bool sparse_enabled = false;
        if (opt_sparse) {
            if (int e = refresh_sparse()) return e;
            sparse_enabled = n_active_min > 0 && n_active_max * 2 <= g.Pw;
        } else if (st->mraf_enabled && st->method != HGS_GS && opt_mraf_split && opt_gh2_mask && tile_geometry_ok() && g.Pw >= 4096) {
            // single-pass MRAF: which columns hold a NaN target (the noise part exists only there) -- a fact about the
            // target, scanned once per upload; the dense launches themselves still walk every column
            if (int e = refresh_sparse()) return e;
        } else if (sizeof(R) == 4 && g.Ph == 4096 && B == 1 && opt_tile2) {
            // dense launches of one hologram at 4096 rows: how many columns hold anything picks the instance of the half-width
            // tile kernel (ColArgs::few_active) -- a fact about the target, scanned once per upload
            if (int e = refresh_sparse()) return e;
        }
        // "computational_spot" statistics on the sparse path: amp_ff is produced on the spot columns dilated
        // by the integration window (col_kernel FWD|STORE over that list) before the fused kernel runs
        const bool spot_stats = stat_ctx && (stat_ctx->groups & 2);
        if (sparse_enabled && spot_stats) {
            // windows sit at floor(spot_knm) (analysis.take, quirk A18), the weights at rint(spot_knm): one more
            // column to the left
            const int lo = (int)std::floor(-(stat_ctx->width - 1) / 2.0);
            if (int e = refresh_dilated(lo - 1, lo + stat_ctx->width - 1)) return e;
            if (int e = need_ff()) return e;
        }
        const int store_sparse = spot_stats ? 2 : 1;
        Plan p = plan_iteration(st, hist ? hist : nullptr);
        const bool sp = sparse_enabled;
        if (int e = keep_prev_phase(p, n)) return e;
        // (the previous call may have left G of these columns behind: gh_state, row_kernel MODE 3)
        if (!gh_holds(sp ? store_sparse : 0)) { if (int e = run_row(0, false, 0, sp ? store_sparse : 0)) return e; }
        for (int i = 0; i < n; ++i) {
            gh_state = -1;          // the column pass turns G into H in place
            if (p.use_fixed || p.store_phase) { if (int e = need_pff()) return e; }
            // MRAF with a weight update takes two passes over the columns: the rebuilt field mixes the
            // NORMALISED weights (signal region) with the un-weighted farfield (noise region), so ||w'|| has
            // to be known first.  Pass 0: forward transform + weight update (+ statistics), no inverse;
            // then wscale = 1/||w'||; pass 1: forward transform again, rebuild, inverse.
            const bool two_pass = st->mraf_enabled && p.do_update;
            // ... unless the tile-resident kernel runs the column pass: the inverse transform is linear, so it transforms the
            // signal part (un-normalised new weights) and the noise part separately in ONE pass and the row kernel joins
            // them once ||w'|| is known (col_tile_kernel RULE 3, row_kernel SPLIT)
            const int m0 = tile_shift(), m1 = m0 + tile_slots() - 1;   // (m0: row shift; m1 - m0 + 1 = slots of the load layout the SLM rows occupy)
            const bool tile_path = (!sp || sparse_tiles) && tile_geometry_ok();
            // (a column list rounded to whole tiles: the same kernels walk the list)
            const int tile_grid = sp ? std::max(1, std::min(tile_blocks, n_active_max / 4)) : tile_blocks;
            const bool split = two_pass && tile_path && g.Pw >= 4096 && opt_mraf_split;
            // ... and the float64 per-column kernel the same way, its noise part through a farfield buffer and an inverse-only
            // launch over the columns that hold noise (CParams::split): one forward transform and one read of weights and
            // target per column instead of two
            // (float32 too where the tile-resident kernel does not run: SLM rows over more than six register slots, short columns)
            const bool split64_ok = two_pass && !tile_path && g.Pw >= 4096 && opt_mraf_split && opt_mraf_split64;
            // (a column list: only where at most half of the listed columns hold noise -- where every one does, as around a noise
            //  box, the single pass saves no transform and pays the extra launch: measured 119 against 105 us at 4096^2)
            // (not where the single-inverse form below takes the update: presum_ok)
            const bool presum_ok = two_pass && opt_mraf_presum && w_unit && !stat_ctx &&
                                   (st->method == HGS_WGS_LEONARDO || st->method == HGS_WGS_KIM);
            bool split64 = split64_ok && !presum_ok;
            if (split64) {
                if (int e = refresh_noise()) return e;
                if (sp && n_noise_max * 2 > n_active_max) split64 = false;
            }
            if (split64 && !ffb_zeroed) {
                // (a farfield-sized buffer more: 268 MB per float64 hologram at 4096^2.  Where the device cannot give it the
                //  update runs in two passes as it did before round 4 -- slower, same results -- instead of failing the call)
                if (!ffb) {
                    void* p = nullptr;
                    if (hipMalloc(&p, (size_t)B * g.Ph * g.Pw * sizeof(C)) != hipSuccess) { (void)hipGetLastError(); split64 = false; }
                    else ffb = static_cast<C*>(p);
                }
                if (split64) {
                    HIPCHK(hipMemsetAsync(ffb, 0, (size_t)B * g.Ph * g.Pw * sizeof(C), stream));
                    ffb_zeroed = true;
                }
            }
            // ... and with ONE inverse per column where ||w'|| can be had BEFORE the field is rebuilt (round 6): the weights that
            // enter this update are normalised (w_unit), so ||w'||^2 = 1 + D, D = sum over the signal pixels of w'^2 - w^2, which a
            // forward-only pre-pass over the columns that hold signal pixels forms (col_presum_kernel; a quarter of the columns
            // at cfg 5).  The main pass (col_tile_kernel RULE 5) rebuilds with the final scale: no second inverse in the noise
            // columns, no noise part parked in LDS, nothing for the row kernel to join.  WGS-Leonardo / WGS-Kim without in-pass
            // statistics; the first update after new weights or a new target (and every other rule) takes the split form.
            const bool presum = presum_ok && split && sizeof(R) == 4;
            // ... and everywhere else the fused path runs an MRAF update (float64; float32 geometries outside the tile-resident
            // kernel's or narrower than 4096 columns): the per-column kernel makes the pre-pass over the list of signal columns
            // (CParams::presum) and the main pass -- per-column or the generic tile kernel -- rebuilds with the pre-summed scale.
            // Replaces the float64 split form (pass + inverse-only launch over the noise columns + joining row launch) and the
            // two-pass form.
            bool presum_col = presum_ok && !presum;
            if (presum_col) {
                if (int e = refresh_signal()) return e;
                if (n_signal_max <= 0) presum_col = false;
                else if (!dpartial) { if (dalloc(&dpartial, (size_t)B * std::max(std::max(col_blocks, tile_blocks), n_cu * 3))) return HGS_ERR_DEVICE; }
            }
            // (a target without a single finite non-zero pixel: D = 0 trivially, but nothing to gain either -- two plain passes)
            // (the pre-pass' grid: one workgroup per CU slot it can hold)
            const int presum_blocks = std::max(1, std::min(g.Pw / 4, (env_presum_blocks > 0 ? std::min(env_presum_blocks, 3 * n_cu)
                                                                         : (g.Ph >= 8192 ? 1 : 2) * n_cu) / B));
            if (presum) {
                if (int e = refresh_sparse()) return e;         // the column flags (clean unless the weights / target moved)
                if (!dpartial) { if (dalloc(&dpartial, (size_t)B * std::max(std::max(col_blocks, tile_blocks), n_cu * 3))) return HGS_ERR_DEVICE; }
            }
            const bool split_any = split || split64;
            if (split_any && !presum && !gh2) { if (dalloc(&gh2, (size_t)B * g.Sh * g.Pw)) return HGS_ERR_DEVICE; }
            // WGS-Nogrette needs nanmean(feedback / target) over the whole farfield before the update (:1851):
            // one more forward-only pass that just accumulates it
            const bool nog = st->method == HGS_WGS_NOGRETTE && p.do_update;
            if (nog && !nog_dev) { if (dalloc(&nog_dev, (size_t)B)) return HGS_ERR_DEVICE; }
            int r = 0;
            if (sp && spot_stats) {
                r = timed(HGS_K_COL_FWD, [&]() -> int {
                    ColArgs<R> a = col_args();
                    a.col_list = col_list_d;
                    a.n_active = n_active_d_dev;
                    LCHK(launch_col<R>(g.Ph, C_FWD | C_STORE, dim3(list_blocks(n_active_d_max), B), stream, a));
                    return 0;
                });
                if (r) return r;
            }
            if (presum) {                 // the pre-pass (its own profile slot: a forward-only column launch)
                r = timed(HGS_K_COL_FWD, [&]() -> int {
                    ColArgs<R> pa = col_args();
                    pa.cp = cparams(st, p);
                    pa.col_flags = col_active;
                    pa.sig_rows = (g.lane_T > 0 && opt_presum_rows) ? sig_rows : nullptr;
                    pa.wpartial = dpartial;
                    LCHK(presum_launch(g.Ph, m1 - m0 + 1, dim3(presum_blocks, B), stream, pa, m0));
                    return 0;
                });
                if (r) return r;
            }
            int presum_col_blocks = 0;
            if (presum_col) {             // the per-column pre-pass over the signal columns
                r = timed(HGS_K_COL_FWD, [&]() -> int {
                    ColArgs<R> pa = col_args();
                    pa.cp = cparams(st, p);
                    pa.cp.weights_only = 1;
                    pa.cp.presum = 1;
                    pa.col_list = col_list_signal;
                    pa.n_active = n_signal_dev;
                    pa.wpartial = dpartial;
                    // (no more workgroups than the dense launch of this geometry keeps resident: at 8192 rows in float64 one per CU --
                    //  the list over three rounds of workgroups cost the pre-pass a prologue per round)
                    presum_col_blocks = std::max(1, std::min(list_blocks(n_signal_max), env_presum_blocks > 0 ? env_presum_blocks : col_blocks));
                    // (fewer than four columns per workgroup pass: the groups of a 4-column run of the list on one XCD, as for
                    //  the column-list launches of the loop -- the signal columns of an image fill their tiles)
                    const int gp = 8 * (4 / col_cpar());
                    if (list_xmap && col_cpar() < 4 && presum_col_blocks >= gp) {
                        presum_col_blocks -= presum_col_blocks % gp;
                        pa.list_xmap = 1;
                    }
                    LCHK(launch_fused<R>(g.Ph, 0, dim3(presum_col_blocks, B), stream, pa));
                    return 0;
                });
                if (r) return r;
            }
            for (int pass = nog ? -1 : 0; pass < (two_pass && !split_any && !presum_col ? 2 : 1) && !r; ++pass) {
                r = timed(HGS_K_COL_FUSED, [&]() -> int {
                    ColArgs<R> a = col_args();
                    a.cp = cparams(st, p);
                    int phase_mode = p.use_fixed ? 2 : (p.store_phase ? 1 : 0);
                    if (pass == -1) {                 // Nogrette: sum of fc only
                        a.cp.nog_pass = 1;
                        a.cp.weights_only = 1;
                        phase_mode = 0;
                    } else if (nog) {
                        a.cp.nog = nog_dev;
                    }
                    if (two_pass && !split_any && !presum_col && pass == 0) {
                        a.cp.weights_only = 1;
                        phase_mode = 0;
                    }
                    if (presum_col) {
                        a.dpartial = dpartial;
                        a.n_dpartial = presum_col_blocks;
                    }
                    if (split64 && pass == 0) {
                        a.cp.split = 1;
                        a.ffb = ffb;
                        row_split = row_split_noise_only = true;
                    }
                    if (two_pass && pass == 1) a.cp.do_update = 0;
                    if (stat_ctx && pass == 0 && (!sp || (stat_ctx->groups & 1))) {
                        a.do_stats = sp ? (stat_ctx->groups & 1) : stat_ctx->groups;   // sparse: amp_ff already stored
                        a.spartial = stat_partial;
                        a.tsum = stat_tsum;
                        a.inv_fsum = 1.0 / amp_norm2;
                        // launches of different geometry share the partial buffer: reset the slots
                        hipLaunchKernelGGL(stat_fill_neutral, dim3((unsigned)((stat_nslots + 255) / 256)), dim3(256), 0, stream,
                                           stat_partial, stat_nslots);
                    }
                    wpartial_n = col_blocks;
                    if (sp) {
                        a.col_list = col_list;
                        a.n_active = n_active_dev;
                    }
                    if (sp && !tile_path) {
                        int blocks = list_blocks(n_active_max);
                        // fewer than four columns per workgroup pass: the groups of a 4-column run of the list on one XCD
                        // (they share 32-byte tile rows where the active set is dense; ColArgs::list_xmap)
                        const int gp = 8 * (4 / col_cpar());
                        if (list_xmap && col_cpar() < 4 && blocks >= gp) {
                            blocks -= blocks % gp;
                            a.list_xmap = 1;
                        }
                        wpartial_n = blocks;
                        if (a.do_stats) LCHK(launch_fused_stats<R>(g.Ph, phase_mode, dim3(blocks, B), stream, a));
                        else LCHK(fused_launch(phase_mode, dim3(blocks, B), a));
                    } else if (presum && pass == 0) {
                        wpartial_n = tile_grid;
                        a.dpartial = dpartial;
                        a.n_dpartial = presum_blocks;
                        if (sp) a.col_flags = col_active;
                        LCHK(tile_presum(g.Ph, phase_mode, m1 - m0 + 1, dim3(tile_grid, B), stream, a, m0));
                    } else if (split && pass == 0) {
                        wpartial_n = tile_grid;
                        a.gh2 = gh2;
                        a.gh2_sparse = (opt_gh2_mask && !sparse_dirty && !sp) ? 1 : 0;   // (set together with RowArgs::gh2_mask below)
                        if (sp) a.col_flags = col_active;       // (scanned with the weights / target this loop started from)
                        LCHK(tile_split(g.Ph, phase_mode, m1 - m0 + 1, opt_tile_rule, dim3(tile_grid, B), stream, a, m0));
                        row_split = true;
                    } else if (int t2 = tile2_grid(sp, tile_path, a, phase_mode)) {
                        // half-width tile-resident kernel: batches at 4096 rows (three workgroups per CU), dense launches at
                        // 2048 rows (col_tile2_kernel); plain passes only
                        wpartial_n = t2;
                        a.few_active = (!sparse_dirty && n_active_min > 0 && n_active_max * 4 <= g.Pw) ? 1 : 0;
                        LCHK(tile2_launch(g.Ph, phase_mode, a.cp.do_update ? 1 : 2, m1 - m0 + 1, dim3(t2, B), stream, a, m0,
                                          (g.Ph >= 4096 && t2 % 16 == 0) ? 1 : 0));
                    } else if (tile_path) {
                        wpartial_n = tile_grid;
                        const bool extras = a.cp.mraf || a.cp.nog_pass || a.cp.weights_only;
                        // MRAF without a weight update (GS, iteration 0, the no-update bodies of WGS): the rule-free MRAF form
                        // compiled per slot count (col_tile_kernel RULE 6) instead of the generic six-slot instance
                        const bool mraf_plain = a.cp.mraf && !a.cp.do_update && !a.cp.nog_pass && !a.cp.weights_only &&
                                                !a.do_stats && opt_tile_rule && opt_mraf_presum && sizeof(R) == 4 && !a.cp.zero_mode;
                        if (mraf_plain) {
                            if (sp) a.col_flags = col_active;
                            LCHK(tile_presum(g.Ph, phase_mode, m1 - m0 + 1, dim3(tile_grid, B), stream, a, m0));
                        } else
                        if (extras) {
                            if (a.do_stats) LCHK(launch_tile_extras_stats<R>(g.Ph, phase_mode, dim3(tile_grid, B), stream, a, m0));
                            else LCHK(launch_tile_extras<R>(g.Ph, phase_mode, dim3(tile_grid, B), stream, a, m0));
                        } else {
                            // the hot launches: weight rule compiled in (col_tile_kernel RULE) where it is the
                            // Leonardo / Kim update or no update at all
                            const int rule = !opt_tile_rule ? 0 : !a.cp.do_update ? 2
                                             : (a.cp.method == HGS_WGS_LEONARDO || a.cp.method == HGS_WGS_KIM) ? 1 : 0;
                            if (a.do_stats) LCHK(launch_tile_stats<R>(g.Ph, phase_mode, dim3(tile_grid, B), stream, a, m0));
                            else if (rule != 0) LCHK(tile_rule(g.Ph, phase_mode, rule, opt_tile_nr4 ? m1 - m0 + 1 : 6, dim3(tile_grid, B), stream, a, m0));
                            else LCHK(launch_tile<R>(g.Ph, phase_mode, dim3(tile_grid, B), stream, a, m0));
                        }
                    } else {
                        if (a.do_stats) LCHK(launch_fused_stats<R>(g.Ph, phase_mode, dim3(col_blocks, B), stream, a));
                        else LCHK(fused_launch(phase_mode, dim3(col_blocks, B), a));
                    }
                    if (pass == -1) {
                        if (int e = reduce(wpartial, wpartial_n, sums + 1 * B)) return e;
                        hipLaunchKernelGGL(nog_finalize<R>, dim3((B + 63) / 64), dim3(64), 0, stream, (const double*)(sums + 1 * B),
                                           sp ? (const int*)n_active_dev : (const int*)nullptr, g.Ph, g.Pw, nog_dev, B);
                        HIPCHK(hipGetLastError());
                    }
                    // (the single-inverse pass needs wscale only from the NEXT column launch on: the row launch below folds the
                    //  partials, as after a plain update -- one 4.7 us launch less per iteration)
                    if (two_pass && pass == 0 && !presum && !presum_col) {
                        hipLaunchKernelGGL(reduce_to_scale<R>, dim3(B), dim3(256), 0, stream, (const double*)wpartial, wpartial_n,
                                           sums + 2 * B, wscale);
                        HIPCHK(hipGetLastError());
                    }
                    return 0;
                });
            }
            if (r) return r;
            if (split64 && n_noise_max > 0) {        // the noise part: farfield values -> gh2, noise columns only (own profile slot)
                r = timed(HGS_K_COL_INV, [&]() -> int {
                    ColArgs<R> nb = col_args();
                    nb.ff = ffb;
                    nb.gh = gh2;
                    nb.col_list = col_list_noise;
                    nb.n_active = n_noise_dev;
                    LCHK(launch_col<R>(g.Ph, C_LOAD | C_INV, dim3(list_blocks(n_noise_max), B), stream, nb));
                    return 0;
                });
                if (r) return r;
            }
            if (stat_ctx) { if (int e = fused_stats_finish(i)) return e; }
            if (p.store_phase) have_pff = true;
            if (p.do_update) { w_pending = true; w_unit = true; }       // (wscale: from this pass' partials, here or in the row launch below)
            if (stat_ctx) { if (int e = eff_gate_after(st, nullptr, stat_ctx->dev_out + ((size_t)i * 2 + st->efficiency_group) * B * 4)) return e; }
            st->iter++;
            Plan pn{0, 0, 0};
            if (i + 1 < n) pn = plan_iteration(st, hist ? hist + i + 1 : nullptr);
            // the row kernel that follows folds the weight-norm partials into wscale (unless already done);
            // on the sparse path it reads the active columns and writes those the next column launches read
            // the last launch of the call extracts the phase; in float32 (and unless a single-pass MRAF body has to join its
            // two parts) it also leaves G of the next body behind (MODE 3) -- of EVERY column, also on the column-list path,
            // so that whatever comes next (another call, the transform that ends optimize()) can start from it on every path
            const int last_mode = (opt_keep_g && sizeof(R) == 4 && !row_split) ? 3 : 1;
            const bool last = i + 1 == n;
            if (int e = run_row(last ? last_mode : 2, p.do_update != 0 && (!two_pass || presum || presum_col), sp ? 1 : 0, (last && last_mode == 3) ? 0 : (sp ? store_sparse : 0)))
                return e;
            p = pn;
        }
        return 0;
    }

    // ---- hgs_iterate_stats: the loop of optimize_gs with stat_groups (_hologram.py:1465-1490) -------------
    // Fused path: the column kernel accumulates the "computational" statistics of the field it
    // constrains (StatAcc) and, for the spot group, stores amp_ff for the window sums; one tiny
    // finalize launch per iteration; the host reads all n_iter results once at the end.
    int fused_stats_finish(int i) {
        StatCtx& c = *stat_ctx;
        const int nparts = wpartial_n * STAT_WAVES;
        if (c.groups & 1) {
            hipLaunchKernelGGL(stat_finalize, dim3(B), dim3(256), 0, stream, (const double*)stat_partial, nparts,
                               (const double*)stat_tsum, 1.0 / amp_norm2, c.dev_out + ((size_t)i * 2 + 0) * B * 4);
            HIPCHK(hipGetLastError());
        }
        if (c.groups & 2) {
            SpotArgs<R> sa{};
            sa.g = g; sa.n_spots = cfg.n_spots; sa.width = c.width; sa.feedback = 1; sa.spot_xy = c.dxy; sa.amp_ff = aff;
            sa.fb = spot_fb;
            hipLaunchKernelGGL(spot_window<R>, dim3((cfg.n_spots + 127) / 128, B), dim3(128), 0, stream, sa);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(spot_stat_finalize<R>, dim3(B), dim3(256), 0, stream, (const R*)spot_fb,
                               (const double*)spot_amp, cfg.n_spots, amp_norm2, c.dev_out + ((size_t)i * 2 + 1) * B * 4);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }

    int iterate_stats(hgs_step* st, int n, uint8_t* hist, int groups, int width, const double* xy,
                      double* out) override {
        if (n < 0) return fail(HGS_ERR_ARG, "n_iter must be >= 0");
        if (groups & ~3) return fail(HGS_ERR_ARG, "unknown statistics group mask %d", groups);
        if (groups == 0) return iterate(st, n, hist);
        if (!out) return fail(HGS_ERR_ARG, "statistics requested without an output buffer");
        if (n == 0) return 0;
        if (cfg.kind != 0) return fail(HGS_ERR_UNSUPPORTED, "hgs_iterate_stats is for the padded-grid holograms");
        if (int e = check_step(st)) return e;
        if (int e = eff_gate_check(st, groups)) return e;
        std::vector<int32_t> ixy;
        if (groups & 2) {
            if (cfg.n_spots <= 0 || !xy || !has_spots) return fail(HGS_ERR_STATE, "spot statistics need spots");
            const int N = cfg.n_spots;
            ixy.resize(2 * N);
            const int flo = (int)std::floor(-(width - 1) / 2.0), fhi = flo + width - 1;
            for (int k = 0; k < N; ++k) {
                ixy[k] = (int32_t)std::floor(xy[k]);
                ixy[N + k] = (int32_t)std::floor(xy[N + k]);
                if (ixy[k] + flo < 0 || ixy[N + k] + flo < 0 || ixy[k] + fhi >= g.Pw || ixy[N + k] + fhi >= g.Ph)
                    return fail(HGS_ERR_ARG, "integration window of spot %d leaves the grid", k);
            }
        }
        for (size_t k = 0; k < (size_t)n * 2 * B * 4; ++k) out[k] = NAN;
        const bool fused = (fused_ok(st) || spot_sparse_ok(st)) && !opt_stepwise;
        if (!fused) {
            have_prev = false;
            // general path: materialise, reduce, constrain -- one host read of a few doubles per iteration
            for (int i = 0; i < n; ++i) {
                if (int e = n2f(0)) return e;
                if (groups & 1) { if (int e = stats(0, 1, nullptr, out + ((size_t)i * 2 + 0) * B * 4)) return e; }
                if (groups & 2) { if (int e = stats(1, width, xy, out + ((size_t)i * 2 + 1) * B * 4)) return e; }
                Plan p = plan_iteration(st, hist ? hist + i : nullptr);
                if (int e = constraint_planned(st, p)) return e;
                if (int e = f2n()) return e;
                if (int e = eff_gate_after(st, out + ((size_t)i * 2 + st->efficiency_group) * B * 4, nullptr)) return e;
                st->iter++;
            }
            return 0;
        }
        StatCtx c;
        c.groups = groups;
        c.width = width;
        const int max_blocks = std::max(std::max(tile_blocks, col_blocks), n_cu * 3);
        const size_t nslots = (size_t)B * max_blocks * STAT_WAVES;
        stat_nslots = nslots;
        if (!stat_partial) { if (dalloc(&stat_partial, nslots * STAT_N)) return HGS_ERR_DEVICE; }
        if (!stat_tsum) { if (dalloc(&stat_tsum, (size_t)B)) return HGS_ERR_DEVICE; }
        if (groups & 2) { if (int e = need_aff()) return e; }
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&c.dev_out), (size_t)n * 2 * B * 4 * sizeof(double)));
        int r = 0;
        do {
            if (groups & 2) {
                if (hipMalloc(reinterpret_cast<void**>(&c.dxy), ixy.size() * sizeof(int)) != hipSuccess) { r = fail(HGS_ERR_DEVICE, "hipMalloc"); break; }
                if (hipMemcpyAsync(c.dxy, ixy.data(), ixy.size() * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) { r = fail(HGS_ERR_DEVICE, "hipMemcpy"); break; }
            }
            hipLaunchKernelGGL(stat_fill_neutral, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, stream, stat_partial, nslots);
            // sum T^2 (the fused path has no NaN targets)
            hipLaunchKernelGGL(ew_sumsq<R>, dim3(ew_blocks, B), dim3(256), 0, stream, (const R*)t, P, epartial);
            if (hipGetLastError() != hipSuccess) { r = fail(HGS_ERR_DEVICE, "statistics setup launch failed"); break; }
            if ((r = reduce(epartial, ew_blocks, stat_tsum))) break;
            stat_ctx = &c;
            r = iterate(st, n, hist);
            stat_ctx = nullptr;
            if (r) break;
            std::vector<double> h((size_t)n * 2 * B * 4);
            if (hipMemcpyAsync(h.data(), c.dev_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) { r = fail(HGS_ERR_DEVICE, "statistics read-back failed"); break; }
            for (int i = 0; i < n; ++i)
                for (int gidx = 0; gidx < 2; ++gidx)
                    if (groups & (1 << gidx))
                        std::memcpy(out + ((size_t)i * 2 + gidx) * B * 4, h.data() + ((size_t)i * 2 + gidx) * B * 4, (size_t)B * 4 * sizeof(double));
        } while (0);
        stat_ctx = nullptr;
        hipStreamSynchronize(stream);
        if (c.dev_out) hipFree(c.dev_out);
        if (c.dxy) hipFree(c.dxy);
        return r;
    }

    hipEvent_t timed_ev[2] = {nullptr, nullptr};      // hgs_iterate_timed: created once (a pair per call costs ~10 us)
    int iterate_timed(hgs_step* st, int n, double* ms) override {
        if (!timed_ev[0]) {
            HIPCHK(hipEventCreate(&timed_ev[0]));
            HIPCHK(hipEventCreate(&timed_ev[1]));
        }
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipEventRecord(timed_ev[0], stream));
        int r = iterate(st, n, nullptr);
        HIPCHK(hipEventRecord(timed_ev[1], stream));
        HIPCHK(hipEventSynchronize(timed_ev[1]));
        float f = 0;
        HIPCHK(hipEventElapsedTime(&f, timed_ev[0], timed_ev[1]));
        *ms = f;
        return r;
    }

    // ---- statistics (_stats.py:7-116): device reductions, host finishing on a handful of doubles ----
    static void finish_stats(const std::vector<double>& fvals, const std::vector<double>& tvals, double total,
                             bool has_total, double* out) {
        // host finishing for short vectors (spot groups): direct restatement on doubles
        double sf = 0, st = 0, stf = 0;
        const size_t n = fvals.size();
        for (size_t i = 0; i < n; ++i) {
            sf += fvals[i] * fvals[i];
            if (tvals[i] == tvals[i]) st += tvals[i] * tvals[i];
        }
        double eff;
        if (has_total) eff = sf / total;
        for (size_t i = 0; i < n; ++i)
            if (tvals[i] == tvals[i]) stf += (tvals[i] / std::sqrt(st)) * (fvals[i] / std::sqrt(sf));
        if (!has_total) eff = stf * stf;
        double rmin = INFINITY, rmax = -INFINITY, emin = INFINITY, emax = -INFINITY, es = 0, es2 = 0, cnt = 0;
        for (size_t i = 0; i < n; ++i) {
            const double tp = tvals[i] * tvals[i] / st, fp = fvals[i] * fvals[i] / sf;
            if (tp != 0 && tp == tp) {
                const double ratio = fp / tp, err = tp - fp;
                rmin = std::fmin(rmin, ratio); rmax = std::fmax(rmax, ratio);
                emin = std::fmin(emin, err); emax = std::fmax(emax, err);
                es += err; es2 += err * err; cnt += 1;
            }
        }
        const double mean = es / cnt, var = std::fmax(0.0, es2 / cnt - mean * mean);
        out[0] = eff;
        out[1] = 1 - (rmax - rmin) / (rmax + rmin);
        out[2] = cnt * (emax - emin);
        out[3] = cnt * std::sqrt(var);
    }

    int stats(int group, int width, const double* xy, double* out) override {
        if (!aff || !farfield_valid) return fail(HGS_ERR_STATE, "statistics need a materialised farfield");
        const int nb = std::min(ew_blocks, 1024);
        if (group == 0) {
            // persistent scratch: hipMalloc / hipFree per call would synchronise the device every iteration
            if (!stats_scratch) { if (dalloc(&stats_scratch, (size_t)B * 1024 * 7 + (size_t)B * 2)) return HGS_ERR_DEVICE; }
            double* d1 = stats_scratch;
            double* d_sfst = d1 + (size_t)B * nb * 7;
            hipLaunchKernelGGL(stats_pass1<R>, dim3(nb, B), dim3(256), 0, stream, (const R*)aff, (const R*)t, P, d1);
            HIPCHK(hipGetLastError());
            std::vector<double> h((size_t)B * nb * 7);
            HIPCHK(hipMemcpyAsync(h.data(), d1, (size_t)B * nb * 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            std::vector<double> sfst(2 * B), stf(B);
            for (int b = 0; b < B; ++b) {
                double s0 = 0, s1 = 0, s2 = 0;
                for (int i = 0; i < nb; ++i) {
                    const double* o = &h[((size_t)b * nb + i) * 3];
                    s0 += o[0]; s1 += o[1]; s2 += o[2];
                }
                sfst[2 * b] = s0; sfst[2 * b + 1] = s1; stf[b] = s2;
            }
            HIPCHK(hipMemcpyAsync(d_sfst, sfst.data(), 2 * B * sizeof(double), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(stats_pass2<R>, dim3(nb, B), dim3(256), 0, stream, (const R*)aff, (const R*)t, P,
                               (const double*)d_sfst, d1);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(h.data(), d1, (size_t)B * nb * 7 * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            for (int b = 0; b < B; ++b) {
                double rmin = INFINITY, rmax = -INFINITY, emin = INFINITY, emax = -INFINITY, es = 0, es2 = 0, cnt = 0;
                for (int i = 0; i < nb; ++i) {
                    const double* o = &h[((size_t)b * nb + i) * 7];
                    rmin = std::fmin(rmin, o[0]); rmax = std::fmax(rmax, o[1]);
                    emin = std::fmin(emin, o[2]); emax = std::fmax(emax, o[3]);
                    es += o[4]; es2 += o[5]; cnt += o[6];
                }
                const double eff = stf[b] / std::sqrt(sfst[2 * b] * sfst[2 * b + 1]);
                const double mean = es / cnt, var = std::fmax(0.0, es2 / cnt - mean * mean);
                out[4 * b + 0] = eff * eff;
                out[4 * b + 1] = 1 - (rmax - rmin) / (rmax + rmin);
                out[4 * b + 2] = cnt * (emax - emin);
                out[4 * b + 3] = cnt * std::sqrt(var);
            }
            return 0;
        }
        if (group == 1) {
            if (cfg.n_spots <= 0 || !xy) return fail(HGS_ERR_STATE, "spot statistics need spots");
            const int N = cfg.n_spots;
            // window sums at floor(spot_knm) (take() floors, quirk A18); width 1 = the pixel itself
            std::vector<int32_t> ixy(2 * N);
            const int flo = (int)std::floor(-(width - 1) / 2.0), fhi = flo + width - 1;
            for (int n = 0; n < N; ++n) {
                ixy[n] = (int32_t)std::floor(xy[n]);
                ixy[N + n] = (int32_t)std::floor(xy[N + n]);
                if (ixy[n] + flo < 0 || ixy[N + n] + flo < 0 || ixy[n] + fhi >= g.Pw || ixy[N + n] + fhi >= g.Ph)
                    return fail(HGS_ERR_ARG, "integration window of spot %d leaves the grid", n);
            }
            if (!stats_dxy) { if (dalloc(&stats_dxy, (size_t)2 * N)) return HGS_ERR_DEVICE; }
            int* dxy = stats_dxy;
            HIPCHK(hipMemcpyAsync(dxy, ixy.data(), 2 * N * sizeof(int), hipMemcpyHostToDevice, stream));
            SpotArgs<R> s{};
            s.g = g; s.n_spots = N; s.width = width; s.feedback = 1; s.spot_xy = dxy; s.amp_ff = aff; s.fb = spot_fb;
            hipLaunchKernelGGL(spot_window<R>, dim3((N + 127) / 128, B), dim3(128), 0, stream, s);
            HIPCHK(hipGetLastError());
            std::vector<R> fb((size_t)B * N);
            std::vector<double> tv(N), fs(B);
            HIPCHK(hipMemcpyAsync(fb.data(), spot_fb, (size_t)B * N * sizeof(R), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(tv.data(), spot_amp, N * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(fs.data(), sums, B * sizeof(double), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            for (int b = 0; b < B; ++b) {
                std::vector<double> fv(N);
                for (int n = 0; n < N; ++n) fv[n] = (double)fb[(size_t)b * N + n];
                finish_stats(fv, tv, fs[b], true, out + 4 * b);
            }
            return 0;
        }
        return fail(HGS_ERR_ARG, "unknown statistics group %d", group);
    }

    int set_option(int option, int value) override {
        // (a change of the column policy may change which columns the next launch expects in gh)
        if (option == HGS_OPT_SPARSE_COLUMNS || option == HGS_OPT_FORCE_STEPWISE || option == HGS_OPT_TILE_KERNEL) gh_state = -1;
        switch (option) {
            case HGS_OPT_SPARSE_COLUMNS: opt_sparse = value ? 1 : 0; return 0;
            case HGS_OPT_FORCE_STEPWISE: opt_stepwise = value ? 1 : 0; return 0;
            case HGS_OPT_TILE_KERNEL: opt_tile = value ? 1 : 0; sparse_dirty = true; return 0;
            case HGS_OPT_SEPARABLE: opt_separable = value ? 1 : 0; return 0;
            case HGS_OPT_SEPARABLE_MIN_SPOTS: opt_sep_min = value > 0 ? value : 1; return 0;
            case HGS_OPT_RUN_KERNELS: opt_run = value ? 1 : 0; return 0;
            case HGS_OPT_KEEP_PREV_PHASE: opt_prev_phase = value ? 1 : 0; if (!value) have_prev = false; return 0;
            case HGS_OPT_ROCTX:
                if (value && !g_roctx.load()) return fail(HGS_ERR_UNSUPPORTED, "no roctx library (librocprofiler-sdk-roctx / libroctx64) found");
                opt_roctx = value ? 1 : 0;
                return 0;
        }
        return fail(HGS_ERR_ARG, "unknown option %d", option);
    }
    int sync() override {
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
};

// makes `dev` current for the calling thread, restores the previous device when it goes out of scope
struct DeviceGuard {
    int prev = -1, dev;
    bool ok = true;
    DispatchLog* prev_log;
    explicit DeviceGuard(int d, DispatchLog* log = nullptr) : dev(d), prev_log(g_dispatch) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
        g_dispatch = log;
    }
    ~DeviceGuard() {
        g_dispatch = prev_log;
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

}  // namespace hgs

// =================================================================================================
// C ABI
// =================================================================================================
struct hgs_engine {
    hgs::EngineBase* impl;
};

extern "C" {

int hgs_create(const hgs_config* cfg, hgs_engine** out) {
    if (!cfg || !out) return hgs::fail(HGS_ERR_ARG, "null argument");
    *out = nullptr;
    hgs::EngineBase* impl = nullptr;
    if (cfg->real_bytes == 4) impl = new hgs::Engine<float>();
    else if (cfg->real_bytes == 8) impl = new hgs::Engine<double>();
    else return hgs::fail(HGS_ERR_ARG, "real_bytes must be 4 or 8 (got %d)", cfg->real_bytes);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    int e = impl->init(*cfg);
    if (prev >= 0 && prev != cfg->device) (void)hipSetDevice(prev);
    if (e) {
        delete impl;
        return e;
    }
    *out = new hgs_engine{impl};
    return 0;
}

int hgs_destroy(hgs_engine* e) {
    if (!e) return 0;
    delete e->impl;
    delete e;
    return 0;
}

// every entry point validates the handle and makes the engine's device current for the calling thread
// (lazy allocations, event creation and launches all follow the current device)
// and puts the caller's device back on return: a torch process whose current device differs from the engine's
// keeps allocating and launching where it was
#define ENG(e)                                                                         \
    if (!(e) || !(e)->impl) return hgs::fail(HGS_ERR_ARG, "null engine handle");       \
    hgs::DeviceGuard guard_((e)->impl->device, &(e)->impl->dispatch);                   \
    if (!guard_.ok) return hgs::fail(HGS_ERR_DEVICE, "hipSetDevice(%d) failed", (e)->impl->device);

int hgs_set_array(hgs_engine* e, int which, const void* host, size_t nbytes) { ENG(e) return e->impl->set_array(which, host, nbytes, false); }
int hgs_set_array_device(hgs_engine* e, int which, const void* dev, size_t nbytes) { ENG(e) return e->impl->set_array(which, dev, nbytes, true); }
int hgs_copy_phase(hgs_engine* dst, hgs_engine* src) {
    if (!src || !src->impl) return hgs::fail(HGS_ERR_ARG, "null source engine");
    ENG(dst)
    hgs::EngineBase::PhaseRef ref;
    if (int r = src->impl->phase_ref(&ref)) return r;
    return dst->impl->copy_phase_from(ref);
}
int hgs_get_array(hgs_engine* e, int which, void* host, size_t nbytes) { ENG(e) return e->impl->get_array(which, host, nbytes, false); }
int hgs_get_array_device(hgs_engine* e, int which, void* dev, size_t nbytes) { ENG(e) return e->impl->get_array(which, dev, nbytes, true); }
int hgs_reset_weights(hgs_engine* e) { ENG(e) return e->impl->reset_weights(); }
int hgs_reset(hgs_engine* e) { ENG(e) return e->impl->reset_state(); }
int hgs_set_array_sparse(hgs_engine* e, int which, const int32_t* xy, const void* values, int32_t n) {
    ENG(e)
    return e->impl->set_array_sparse(which, xy, values, n);
}
int hgs_nearfield2farfield(hgs_engine* e, int store_phase_ff) { ENG(e) return e->impl->n2f(store_phase_ff); }
int hgs_farfield_constraint(hgs_engine* e, hgs_step* step) {
    ENG(e)
    if (!step) return hgs::fail(HGS_ERR_ARG, "null step");
    return e->impl->constraint(step);
}
int hgs_farfield2nearfield(hgs_engine* e) { ENG(e) return e->impl->f2n(); }
int hgs_multiplane_farfield2nearfield(hgs_engine* const* children, const double* weights, int n) {
    if (!children || !weights || n < 1) return hgs::fail(HGS_ERR_ARG, "multiplane: need at least one child and its weight");
    if (n > hgs::MP_MAX) return hgs::fail(HGS_ERR_UNSUPPORTED, "multiplane: at most %d children", hgs::MP_MAX);
    hgs::EngineBase::MpInfo info[hgs::MP_MAX];
    for (int k = 0; k < n; ++k) {       // validate the whole family before touching any state
        if (!children[k] || !children[k]->impl) return hgs::fail(HGS_ERR_ARG, "multiplane: null child %d", k);
        if (int r = children[k]->impl->mp_info(&info[k])) return r;
        if (info[k].S != info[0].S || info[k].B != info[0].B || info[k].real_bytes != info[0].real_bytes ||
            info[k].device != info[0].device)
            return hgs::fail(HGS_ERR_ARG, "multiplane: child %d differs in SLM shape, batch, precision or device", k);
    }
    hgs::DeviceGuard guard_(info[0].device);
    if (!guard_.ok) return hgs::fail(HGS_ERR_DEVICE, "hipSetDevice(%d) failed", info[0].device);
    for (int k = 0; k < n; ++k) {
        hgs::g_dispatch = &children[k]->impl->dispatch;
        if (int r = children[k]->impl->f2n_complex()) return r;
        if (int r = children[k]->impl->mp_info(&info[k])) return r;     // the nearfield buffer exists now
    }
    // every child's inverse transform must have landed before child 0's stream reads them
    for (int k = 1; k < n; ++k)
        if (hipStreamSynchronize(info[k].stream) != hipSuccess) return hgs::fail(HGS_ERR_DEVICE, "multiplane: stream sync failed");
    hgs::g_dispatch = nullptr;
    return children[0]->impl->mp_combine(info, weights, n);
}
int hgs_iterate(hgs_engine* e, hgs_step* step, int n_iter, uint8_t* hist) {
    ENG(e)
    if (!step) return hgs::fail(HGS_ERR_ARG, "null step");
    return e->impl->iterate(step, n_iter, hist);
}
int hgs_iterate_stats(hgs_engine* e, hgs_step* step, int n_iter, uint8_t* hist, int stat_groups, int width,
                      const double* spot_xy_float, double* stats_out) {
    ENG(e)
    if (!step) return hgs::fail(HGS_ERR_ARG, "null step");
    return e->impl->iterate_stats(step, n_iter, hist, stat_groups, width, spot_xy_float, stats_out);
}
int hgs_stats(hgs_engine* e, int group, int width, const double* xy, double* out) {
    ENG(e)
    if (!out) return hgs::fail(HGS_ERR_ARG, "null output");
    return e->impl->stats(group, width, xy, out);
}
int hgs_sync(hgs_engine* e) { ENG(e) return e->impl->sync(); }
int hgs_set_option(hgs_engine* e, int option, int value) { ENG(e) return e->impl->set_option(option, value); }
int hgs_profile_enable(hgs_engine* e, int on) { ENG(e) return e->impl->profile_enable(on); }
int hgs_profile_read(hgs_engine* e, double* out) {
    ENG(e)
    if (!out) return hgs::fail(HGS_ERR_ARG, "null output");
    return e->impl->profile_read(out);
}
int hgs_iterate_timed(hgs_engine* e, hgs_step* step, int n_iter, double* ms) {
    ENG(e)
    if (!step || !ms) return hgs::fail(HGS_ERR_ARG, "null argument");
    return e->impl->iterate_timed(step, n_iter, ms);
}

int hgs_dispatch_read(hgs_engine* e, char* buf, size_t nbytes, size_t* needed) {
    if (!e || !e->impl) return hgs::fail(HGS_ERR_ARG, "null engine handle");
    const std::string t = e->impl->dispatch.text();
    if (needed) *needed = t.size() + 1;
    if (!buf || nbytes < t.size() + 1) {
        if (buf && nbytes) buf[0] = 0;
        if (!buf && nbytes == 0 && needed) return 0;      // size query: the record is kept
        return hgs::fail(HGS_ERR_ARG, "dispatch record needs %zu bytes", t.size() + 1);
    }
    std::memcpy(buf, t.c_str(), t.size() + 1);
    e->impl->dispatch.v.clear();
    return 0;
}

const char* hgs_last_error(void) { return hgs::g_err.c_str(); }

const char* hgs_version(void) {
    static std::string v;
    if (v.empty()) {
        v = "hgs 0.1 gfx950";
        int n = 0;
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, 0) == hipSuccess) v += std::string(" ") + p.name + " " + p.gcnArchName;
        } else {
            v += " (no device)";
        }
    }
    return v.c_str();
}

}  // extern "C"
