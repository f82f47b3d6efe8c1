/*
 * hgs.h -- C ABI of the MI355X hologram (Gerchberg-Saxton / weighted-GS) engine.
 *
 * This is the drop-in boundary for slmsuite's optimize() hot path.  The reference has no FFI of
 * its own: the path sits behind the array-module alias `cp` (slmsuite/holography/algorithms/
 * _header.py:15-32) and the overridable per-iteration methods of `Hologram`
 * (slmsuite/holography/algorithms/_hologram.py).  Each entry point below replaces the reference
 * method(s) cited next to it; INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative hgs_status otherwise; the message of the
 *     last error of the calling thread is available from hgs_last_error();
 *   - an engine handle is single-threaded; different handles are independent; every entry point makes the
 *     engine's HIP device current for the duration of the call and restores the caller's current device;
 *   - the engine owns all device memory and one HIP stream; host buffers are caller-owned,
 *     C-contiguous, in the reference's natural layout (row-major, centred zero order), of the
 *     engine's real type (float when real_bytes == 4, double when 8; complex = 2 reals);
 *   - batched arrays are [batch][...]; passing exactly one hologram's bytes broadcasts it;
 *   - calls are asynchronous on the engine stream unless they move data to/from the host.
 *   - there is NO CPU fallback: without a usable gfx950 device hgs_create fails with
 *     HGS_ERR_DEVICE.
 */
#ifndef HGS_H
#define HGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hgs_engine hgs_engine;

typedef enum {
    HGS_OK = 0,
    HGS_ERR_ARG = -1,      /* invalid argument (ValueError on the Python side)            */
    HGS_ERR_DEVICE = -2,   /* no device / HIP runtime error (RuntimeError)                */
    HGS_ERR_STATE = -3,    /* operation not valid in the current state                    */
    HGS_ERR_UNSUPPORTED = -4
} hgs_status;

/* Geometry of one engine.  Hologram.__init__ (_hologram.py:196-439): `shape` -> pad_h/pad_w,
 * `slm_shape` -> slm_h/slm_w, dtype (:391-398) -> real_bytes.  Powers of two in [64, 8192] run the fused
 * kernels; any other shape (the reference only warns about those, :378-384) runs the same operators axis by
 * axis on the workgroup transforms (general operators only, no fused loop): a power-of-two axis up to 16384
 * (float64: 8192) directly, any other length in [2, 8192] (float64: [2, 4096]) through Bluestein's identity.
 * Longer axes return HGS_ERR_UNSUPPORTED.  The SLM block is
 * centred (toolbox.unpad, toolbox/__init__.py:1699-1712). */
typedef struct {
    int32_t device;      /* HIP device ordinal                                          */
    int32_t pad_h, pad_w;
    int32_t slm_h, slm_w;
    int32_t real_bytes;  /* 4 | 8                                                       */
    int32_t batch;       /* independent holograms advanced together        */
    int32_t n_spots;     /* > 0 enables spot-window / external-spot feedback            */
    int32_t kind;        /* 0: padded DFT grid (Hologram / SpotHologram);
                            1: CompressedSpotHologram (_spots.py:178-1018): n_spots free-floating
                               spots with polynomial phase kernels, no padded grid (pad_* ignored) */
    int32_t n_monomials; /* kind 1: monomials x^px y^py of the kernel phase polynomial  */
} hgs_config;

/* ALGORITHM_INDEX (_header.py:72) */
enum { HGS_GS = 0, HGS_WGS_LEONARDO = 1, HGS_WGS_KIM = 2, HGS_WGS_NOGRETTE = 3, HGS_WGS_WU = 4,
       HGS_WGS_TANH = 5 };
/* feedback sources of _update_weights (_hologram.py:1914, _spots.py:1573-1624) */
enum { HGS_FB_PIXEL = 0 /* "computational" */, HGS_FB_SPOT_WINDOW = 1 /* "computational_spot" */,
       HGS_FB_EXTERNAL = 2 /* "external_spot" */ };

/* Per-call flags: the POD image of Hologram.flags (_hologram.py:1370-1410).  Fields marked
 * in/out are advanced by the engine exactly as optimize_gs / _gs_farfield_routines do. */
typedef struct {
    int32_t method;              /* HGS_GS ...                                           */
    int32_t feedback;            /* HGS_FB_*                                             */
    int32_t iter;                /* in/out  self.iter (:1490)                            */
    int32_t fixed_phase;         /* in/out  flags["fixed_phase"] (:1556-1585).  Set while the engine holds
                                    no phase_ff (after hgs_reset), an iteration stores the phase first --
                                    the guard of :1601.  The reference's MRAF branch (:1643) lacks that
                                    guard and raises there; a host that wants the
                                    reference's behaviour refuses the call, as the Python class does    */
    int32_t fix_phase_iteration; /* flags["fix_phase_iteration"]                         */
    int32_t false_run;           /* in/out  trailing count of contiguous False entries in
                                    stats["flags"]["fixed_phase"] (:1574-1577); -1 = history
                                    contains a non-boolean entry (never fixes by iteration)   */
    int32_t mraf_enabled;        /* target contains NaN (_mraf_helper_routines :1498)    */
    int32_t has_mraf_factor;     /* flags["mraf_factor"] is not None                     */
    int32_t zero_mode;           /* flags["zero_factor"] given and != 0 (:1514)          */
    int32_t spot_window;         /* spot_integration_width_knm (_spots.py:1284-1297)     */
    int32_t efficiency_group;    /* statistics group whose efficiency gates WGS-Kim (0 "computational",
                                    1 "computational_spot"): the LAST group of stats["stats"] (:1562-1567) */
    int32_t reserved;            /* keeps the doubles 8-byte aligned; set to 0           */
    double feedback_exponent, feedback_factor, mraf_factor, zero_factor;
    double fix_phase_efficiency; /* flags["fix_phase_efficiency"] (:1560-1569), NaN = None.  WGS-Kim only: the
                                    phase is fixed from the iteration AFTER the first one (iter > 0) whose recorded
                                    efficiency exceeds it.  Needs hgs_iterate_stats with that group (batch = 1);
                                    hgs_iterate / hgs_farfield_constraint return HGS_ERR_ARG when it is set without
                                    statistics ("Must track statistics ..."), like the reference's ValueError. */
} hgs_step;

/* array selectors for hgs_set_array / hgs_get_array */
enum {
    HGS_PHASE = 0,        /* [batch][slm_h][slm_w] real      Hologram.phase              */
    HGS_AMP = 1,          /* [slm_h][slm_w] real, unit L2    Hologram.amp (array form)   */
    HGS_AMP_SCALAR = 2,   /* 1 real                          Hologram.amp (scalar form)  */
    HGS_PROP_KERNEL = 3,  /* [slm_h][slm_w] real             Hologram.propagation_kernel; nbytes = 0: none */
    HGS_TARGET = 4,       /* [batch][pad_h][pad_w] real      Hologram.target (may hold NaN) */
    HGS_WEIGHTS = 5,      /* [batch][pad_h][pad_w] real      Hologram.weights            */
    HGS_PHASE_FF = 6,     /* [batch][pad_h][pad_w] real      Hologram.phase_ff           */
    HGS_FARFIELD = 7,     /* [batch][pad_h][pad_w] complex   Hologram.farfield           */
    HGS_AMP_FF = 8,       /* [batch][pad_h][pad_w] real      Hologram.amp_ff             */
    HGS_SPOT_INDEX = 9,   /* int32 [2][n_spots] (kx row 0, ky row 1)  spot_knm_rounded   */
    HGS_SPOT_AMP = 10,    /* double [n_spots]                SpotHologram.spot_amp       */
    HGS_EXTERNAL_AMP = 11,/* double [n_spots]                external_spot_amp           */
    HGS_ZERO_WEIGHTS = 12,/* [batch][pad_h][pad_w] complex   dense image of zero_weights */
    /* kind 1 (CompressedSpotHologram); TARGET/WEIGHTS/PHASE_FF/FARFIELD/AMP_FF are [batch][n_spots] */
    HGS_XGRID = 13,       /* [slm_h][slm_w] real   slm.grid[0] * zernike scaling (_spots.py:614-618) */
    HGS_YGRID = 14,       /* [slm_h][slm_w] real   slm.grid[1] * zernike scaling                     */
    HGS_MONOMIALS = 15,   /* int32 [n_monomials][2] (px, py)   phase._zernike_get_cantor terms; the pseudo-term
                             (-1, 0) is the vortex plate w * atan2(y, x), w > 0 only (phase.py:1783-1790)     */
    HGS_SPOT_COEFF = 16,  /* real [n_monomials][n_spots]       ... and weights (phase.py:850-920)    */
    HGS_PHASE_PREV = 17   /* get only: [batch][slm_h][slm_w] real -- the phase the last one-iteration hgs_iterate call
                             STARTED from (HGS_OPT_KEEP_PREV_PHASE); HGS_ERR_STATE when none is held */
};

int hgs_create(const hgs_config* cfg, hgs_engine** out);
int hgs_destroy(hgs_engine* e);

/* host -> device / device -> host (synchronous).  Replaces the cp.array(...)/.get() traffic of
 * Hologram.__init__, reset_phase (:536), set_weights (:829), get_phase (:786) ... */
int hgs_set_array(hgs_engine* e, int which, const void* host, size_t nbytes);
int hgs_get_array(hgs_engine* e, int which, void* host, size_t nbytes);
/* device -> caller-provided DEVICE buffer (e.g. a torch tensor handed to RCCL); synchronous */
int hgs_get_array_device(hgs_engine* e, int which, void* dev, size_t nbytes);
/* caller-provided DEVICE buffer -> engine (same layouts and sizes as hgs_set_array): what the reference does when it is
 * handed CuPy arrays (`cp.array(x, copy=False)`, _hologram.py:70-77) -- no host bounce.  Arrays the engine parses on the
 * host (HGS_SPOT_INDEX, HGS_XGRID/YGRID, HGS_MONOMIALS, HGS_SPOT_COEFF, HGS_AMP_SCALAR) return HGS_ERR_UNSUPPORTED. */
int hgs_set_array_device(hgs_engine* e, int which, const void* dev, size_t nbytes);
/* dst.phase <- src.phase on the device (same SLM shape and precision; one source hologram broadcasts over dst's batch;
 * engines on different GPUs use a peer copy).  Hologram.get_farfield (_hologram.py:853-931, the per-frame call of
 * SimulatedCamera) transforms the current phase on another grid without moving it through the host. */
int hgs_copy_phase(hgs_engine* dst, hgs_engine* src);
/* Hologram.reset_weights (:603-614): weights = target with NaN -> 0; zero_weights cleared */
int hgs_reset_weights(hgs_engine* e);
/* Hologram.reset (:442-478) as far as the device is concerned: hgs_reset_weights, and phase_ff / farfield /
 * amp_ff go back to "None" (a later WGS-Kim fix stores a fresh phase; reading them back fails with
 * HGS_ERR_STATE until hgs_nearfield2farfield ran).  Phase, amplitude, target, spots and options stay. */
int hgs_reset(hgs_engine* e);
/* Sparse upload of a farfield-sized real array (HGS_TARGET or HGS_WEIGHTS, kind 0): the array becomes ZERO everywhere
 * except values[k] at pixel (kx = xy[k], ky = xy[n + k]), k < n, later entries winning where pixels repeat (NumPy fancy
 * assignment, _spots.py:1541).  One hologram's list broadcasts over the batch.  This is what
 * SpotHologram._set_target_spots produces without null points: n_spots values instead of pad_h * pad_w.  A NaN
 * background (null points) cannot be expressed here: upload such a target densely with hgs_set_array. */
int hgs_set_array_sparse(hgs_engine* e, int which, const int32_t* xy, const void* values, int32_t n);

/* Hologram._nearfield2farfield (:1038-1056) + _midloop_cleaning (:951-953): fills farfield and
 * amp_ff; with store_phase_ff != 0 also phase_ff = atan2(farfield) (_populate_results :934-949). */
int hgs_nearfield2farfield(hgs_engine* e, int store_phase_ff);
/* Hologram._gs_farfield_routines (:1550-1661) incl. _update_weights (:1914 / _spots.py:1573) and
 * _update_weights_generic (:1786-1879); operates on the materialised farfield. */
int hgs_farfield_constraint(hgs_engine* e, hgs_step* step);
/* Hologram._farfield2nearfield (:1058-1073) + _nearfield_extract (:1026-1036). */
int hgs_farfield2nearfield(hgs_engine* e);
/* n_iter bodies of the optimize_gs loop (:1465-1490) with no callback and no statistics:
 * the fused fast path (the farfield is never materialised when the method allows it).
 * fixed_phase_history[n_iter] (optional) receives flags["fixed_phase"] as _update_stats would
 * have recorded it in each iteration. */
int hgs_iterate(hgs_engine* e, hgs_step* step, int n_iter, uint8_t* fixed_phase_history);
/* _HologramStats._calculate_stats (_stats.py:7-116), efficiency_compensation = False.
 * group 0: "computational" (amp_ff vs target); group 1: "computational_spot"
 * (window sums at spot_knm with total power, _spots.py:1626-1679).
 * out[batch][4] = efficiency, uniformity, pkpk_err, std_err.  Needs a materialised farfield. */
int hgs_stats(hgs_engine* e, int group, int width, const double* spot_xy_float, double* out);

/* hgs_iterate for optimize(..., stat_groups=[...]) (_hologram.py:1479): additionally returns
 * the statistics _update_stats records in every iteration, i.e. those of the farfield the iteration
 * starts from.  stat_groups: bit 0 "computational", bit 1 "computational_spot" (needs width and
 * spot_xy_float as hgs_stats).  stats_out[n_iter][2][batch][4] (group slot 0 / 1; slots of groups
 * not requested are NaN).  On the fused path the column kernel accumulates the reductions in the
 * pass that applies the constraint (no farfield is materialised, one host read at the end);
 * otherwise the call runs the general path with a hgs_stats per iteration. */
int hgs_iterate_stats(hgs_engine* e, hgs_step* step, int n_iter, uint8_t* fixed_phase_history,
                      int stat_groups, int width, const double* spot_xy_float, double* stats_out);

/* MultiplaneHologram._farfield2nearfield (_multiplane.py:255-279): every child runs
 * _farfield2nearfield(extract=False) on its own (constrained) farfield; the children's complex
 * nearfields over the SLM are summed on the device,
 *     nf = sum_k weights[k] * nf_k * exp(-i propagation_kernel_k),   phase = atan2(nf),
 * and the common phase is written into every child's HGS_PHASE (the reference's children share
 * one phase array).  Children may differ in pad shape and kind but must agree in SLM shape,
 * batch, precision and device; n <= 16.  Synchronous. */
int hgs_multiplane_farfield2nearfield(hgs_engine* const* children, const double* weights, int n);

int hgs_sync(hgs_engine* e);

/* Engine options.  HGS_OPT_SPARSE_COLUMNS (default 1): in hgs_iterate / hgs_iterate_stats, when few of
 * the farfield columns hold a non-zero weight or target (spot arrays), transform only those columns
 * and move only them between the two kernels; the phase and the weights are those of the dense path
 * (every other column of the constrained farfield is exactly zero).  Covers pixel feedback and, with
 * the spot columns dilated by the integration window, the spot feedback modes.  While it is active,
 * HGS_PHASE_FF stored by WGS-Kim is refreshed on the active columns only (the rest cannot influence the
 * loop; hgs_nearfield2farfield(store_phase_ff = 1) refreshes every pixel).  0 forces the dense kernels.
 *   Where the tile-resident kernel applies (fp32, pad_h >= 4096) and the active columns fill their 4-column tiles at
 *   least half (images, MRAF noise boxes), the active set is rounded up to whole tiles and that kernel walks the tile
 *   list; results are those of the dense launch bit for bit.
 *   MRAF with a weight update runs ONE column pass with ONE inverse per column (round 6): the weights that enter an update are
 *   normalised, so ||w'||^2 = 1 + D, D = the sum over the signal pixels of w'^2 - w^2, which a forward-only pre-pass over the
 *   columns that hold signal pixels forms before the pass rebuilds the field (WGS-Leonardo / WGS-Kim without in-pass statistics;
 *   float32 and float64).  The first update after new weights or a new target -- and the other rules -- split the rebuilt field
 *   instead (its signal and noise part are transformed separately and joined by the row kernel once ||w'|| is known), or take
 *   two passes where neither form applies.
 * HGS_OPT_FORCE_STEPWISE (default 0): hgs_iterate / hgs_iterate_stats loop the three general operators
 *   (materialised farfield) even where a fused kernel exists -- the reference's own op sequence; used by tests.
 * HGS_OPT_TILE_KERNEL (default 1): use the tile-resident fused column kernels where they apply (fp32: col_tile_kernel at
 *   pad_h >= 4096, the half-width col_tile2_kernel at pad_h = 4096 and 2048); 0 forces the per-column kernel at every size.
 * HGS_OPT_SEPARABLE (default 1), HGS_OPT_SEPARABLE_MIN_SPOTS (default 96): kind 1, run the two transforms as
 *   complex GEMMs on the matrix cores when the basis and the grid factorise; 0 forces the direct kernels.
 * HGS_OPT_RUN_KERNELS (default 1): kind 1, fp32, regular pixel grid and a phase polynomial of degree <= 2 (any basis of
 *   tilts, focus and astigmatisms, separable or not): the direct transforms advance exp(i phi) along runs of 16 pixels by
 *   a two-term recurrence instead of evaluating the polynomial, sin and cos per pixel; 0 forces the per-pixel kernels.
 * Options are per engine and take effect at the next call; nothing is read from the environment after
 * hgs_create (which reads the developer overrides once: the grid sizes HGS_ROW_BLOCKS / HGS_COL_BLOCKS / HGS_TILE_BLOCKS /
 * HGS_TILE2_BLOCKS / HGS_ROW_PREF_BLOCKS, and the A/B switches HGS_ROW_XCD, HGS_COL_XMAP, HGS_ROW_SHIFT, HGS_ROW_SHIFT64, HGS_ROW_PREF,
 * HGS_ROW_PREF_BATCH, HGS_TILE_RULE, HGS_MRAF_SPLIT, HGS_MRAF_SPLIT64, HGS_GH2_MASK, HGS_TILE_LIST, HGS_TILE_SHIFT16, HGS_TILE_NR4,
 * HGS_TILE2, HGS_TILE2_MIN_BATCH, HGS_TILE2_PHASE2, HGS_KEEP_G, HGS_FUSED_SHIFT, HGS_MONO_TAB, HGS_MRAF_PRESUM, HGS_PRESUM_ROWS,
 * HGS_PRESUM_BLOCKS -- all default to the tuned path;
 * HGS_TRACE_INIT=1 prints where hgs_create spends its time).
 *   HGS_KEEP_G (default 1): the last row launch of a float32 hgs_iterate call leaves G of the next body behind, and the next
 *   call -- or hgs_nearfield2farfield -- on an unchanged phase starts from it.  That G is the loop's own un-rounded phasor
 *   amp * nf / |nf|, not exp(i * HGS_PHASE) of the rounded, stored phase: the trailing transform is consistent with the loop
 *   (a loop cut into calls walks bit for bit like one call) rather than bit-identical with a fresh engine given the
 *   downloaded phase; the two agree to float32 rounding (tests/test_gpu_round6.py, 2e-6 on the farfield).
 * HGS_OPT_ROCTX (default 0): roctx ranges (hgs_iterate, hgs_nearfield2farfield, hgs_farfield_constraint,
 *   hgs_farfield2nearfield) for rocprofv3 --marker-trace; the roctx library is dlopen'ed on first use.
 * HGS_OPT_KEEP_PREV_PHASE (default 0): a fused hgs_iterate / hgs_iterate_stats call of ONE iteration that rewrites the
 *   farfield phase (i.e. does not use a fixed one) first copies HGS_PHASE to HGS_PHASE_PREV on the device.  The fused
 *   kernels never materialise Hologram.phase_ff (= atan2 of the farfield the iteration STARTED from, _hologram.py:1583 /
 *   :1602); a caller that runs the loop one iteration per call -- Hologram.optimize(callback=...) -- can rebuild it on
 *   demand from that phase (hgs_nearfield2farfield(store_phase_ff = 1) on a second engine).  Calls that iterate the
 *   general operators keep HGS_PHASE_FF itself up to date and hold no previous phase. */
enum { HGS_OPT_SPARSE_COLUMNS = 1, HGS_OPT_FORCE_STEPWISE = 2, HGS_OPT_TILE_KERNEL = 3, HGS_OPT_SEPARABLE = 4,
       HGS_OPT_SEPARABLE_MIN_SPOTS = 5, HGS_OPT_ROCTX = 6, HGS_OPT_RUN_KERNELS = 7, HGS_OPT_KEEP_PREV_PHASE = 8 };
int hgs_set_option(hgs_engine* e, int option, int value);

/* Timing support for bench.py: per-kernel HIP-event timing on the engine stream. */
enum { HGS_K_ROW = 0, HGS_K_COL_FUSED = 1, HGS_K_COL_FWD = 2, HGS_K_COL_INV = 3, HGS_K_ELEMENTWISE = 4,
       HGS_K_COUNT = 5 };
int hgs_profile_enable(hgs_engine* e, int on);
/* out[HGS_K_COUNT][2] = {total milliseconds, launches} since the last call; syncs the stream */
int hgs_profile_read(hgs_engine* e, double* out);
/* elapsed milliseconds of hgs_iterate(step, n_iter) measured with a HIP event pair on the engine
 * stream (SURVEY 8d timing protocol) */
int hgs_iterate_timed(hgs_engine* e, hgs_step* step, int n_iter, double* ms);

/* Dispatch record: which kernel template instance each launch of this engine went to since the previous call, one line
 * per (instance, run-time flags):
 *     "col_tile_kernel<R=float,N=4096,PHASE=0,NR=6,STATS=false,EXTRAS=false,RULE=1,LISTED=0> xmap\t50\n"
 * (family, its template arguments by name -- the ones rocprofv3 prints positionally --, flags list / load_mask /
 * store_mask / xmap / batch / stats / nf_out, launch count).  Families: row_kernel, col_kernel, col_fused_kernel,
 * col_tile_kernel, bluestein_lines, c_n2f_run, c_f2n_run, c_n2f_partial, c_f2n, cgemm_streamk (the small element-wise /
 * reduction helpers are not recorded).  The tests assert it next to the numbers: neighbouring variants often agree to
 * the last bit, so only this shows that a policy reached the kernel it names.  `needed` (optional) receives the bytes
 * the text takes including the terminator; with buf = NULL and nbytes = 0 the call is a size query and keeps the record,
 * otherwise a buffer that is too small fails with HGS_ERR_ARG (record kept) and success clears the record. */
int hgs_dispatch_read(hgs_engine* e, char* buf, size_t nbytes, size_t* needed);

const char* hgs_last_error(void);
/* library / device identification: "hgs <version> gfx950 <device name>" */
const char* hgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HGS_H */
