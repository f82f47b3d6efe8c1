"""
Byte models of the DFT-grid workloads: what one launch of the fused column kernel and of the row kernel
MOVES across the memory system -- the basis of ``roofline.achieved`` -- next to SURVEY 8(d)'s canonical (un-pruned) count.
Pure functions of the geometry, importable without a GPU; tests/test_bench_models.py holds them to the figures and the
PMC-measured traffic recorded in profiles/r04/configs.jsonl.
"""
import os


def tile_slots(Ph, Sh):
    """Register slots of the column kernels' load layout (T = Ph / 16 rows each) that the SLM rows occupy once the transform
    input is shifted by r0 rounded down to a multiple of 16 rows (col_tile_kernel, round 5)."""
    T = Ph // 16
    r0 = (Ph - Sh) // 2
    return (r0 % 16 + Sh + T - 1) // T


def canonical_bytes(P, S, r, wgs, kim_fixed=False):
    """13 P r + 2 S r (GS), 15 P r + 2 S r (WGS), 16 P r + 2 S r (WGS-Kim with a fixed phase) per iteration."""
    return ((16 if kim_fixed else 15 if wgs else 13) * P + 2 * S) * r


def grid_bytes_models(shape, slm, dtype, batch, streams, method, sparse_target, n_targets, mraf, signal_cols=None,
                      noise_cols=None, noise_pixels=None, env=None):
    """
    shape / slm: padded and SLM shape; dtype "f32" | "f64"; batch holograms over ``streams`` stream groups (a launch covers
    one group); sparse_target: spot-like target whose weights change in n_targets lanes only; mraf: the target holds NaN;
    signal_cols / noise_cols / noise_pixels: columns with a finite non-zero / a NaN target and the NaN pixels (cfg 5).
    Returns the dictionary bench.py reports from (bytes per LAUNCH of one stream group, and their description).
    """
    env = os.environ if env is None else env
    Ph, Pw = shape
    Sh, Sw = slm
    P, S = Ph * Pw, Sh * Sw
    r = 4 if dtype == "f32" else 8
    c = 2 * r
    B = -(-batch // streams)      # holograms per launch (one stream group)
    m = method
    wgs = m != "GS"
    gh = Sh * Pw * c                              # half-transformed field: SLM rows only
    # weights are written back per lane (its 16 values of a column, 64 contiguous bytes) where one of them changed:
    # every lane of a column that holds a finite non-zero target, nothing elsewhere
    w_write = (P * r) if not sparse_target else n_targets * 16 * r
    if signal_cols:
        w_write = signal_cols * Ph * r
    kim = m == "WGS-Kim"                          # after the fixing iteration the stored phase_ff is read back
    col = 2 * gh + P * r + (P * r if (wgs or mraf) else 0) + (w_write if wgs else 0) + (P * r if kim else 0)
    passes = 1
    other = 0                                     # launches of an iteration besides the column pass(es) and the row launch
    row = 2 * gh                                  # MODE 2 (between fused iterations): H read, G written
    mraf_note = ""
    other_kind, other_name = "col_inv", "col_kernel<double, N, LOAD | INV> over the columns that hold a NaN target"
    if mraf and wgs:
        slots = tile_slots(Ph, Sh)
        if (dtype == "f32" and Ph >= 4096 and Pw >= 4096 and slots <= 6 and env.get("HGS_MRAF_SPLIT", "1") != "0"
                and env.get("HGS_MRAF_PRESUM", "1") != "0" and m in ("WGS-Leonardo", "WGS-Kim") and signal_cols):
            # round 6: ||w'|| known BEFORE the field is rebuilt.  A forward-only pre-pass over the tiles that hold a signal
            # column (col_presum_kernel: reads their GH rows, weights and targets, writes a partial per workgroup), then ONE
            # column pass with one inverse per column (col_tile_kernel RULE 5: reads GH, w, t; writes w and GH), plain row launch.
            # (the steady state of a loop; the first update after new weights takes the split form below)
            other = gh * signal_cols // Pw + 2 * signal_cols * Ph * r
            other_kind, other_name = "col_fwd", "col_presum_kernel (forward-only pre-pass over the columns that hold signal pixels)"
            mraf_note = ("; MRAF with a weight update and ONE inverse per column: 1 / ||w'|| = 1 / sqrt(1 + D) from a forward-only "
                         f"pre-pass over the {signal_cols} columns that hold signal pixels (its own launch, reported beside this one)")
        elif env.get("HGS_MRAF_PRESUM", "1") != "0" and m in ("WGS-Leonardo", "WGS-Kim") and signal_cols:
            # the same update on the per-column kernel (float64; float32 outside the tile-resident geometry): the pre-pass is a
            # forward-only launch of col_fused_kernel over the list of signal columns (CParams::presum), the main pass one plain
            # launch with one inverse per column, the row launch the plain one
            if dtype == "f64":
                w_write = n_targets * r            # (float64 leaves changed weights four pixels = 32 bytes at a time)
                col = 2 * gh + 2 * P * r + w_write
            other = gh * signal_cols // Pw + 2 * signal_cols * Ph * r
            other_kind, other_name = "col_fwd", "col_fused_kernel over the list of signal columns, forward only (CParams::presum)"
            mraf_note = ("; MRAF with a weight update and ONE inverse per column: 1 / ||w'|| = 1 / sqrt(1 + D) from a forward-only "
                         f"pre-pass over the {signal_cols} columns that hold signal pixels (its own launch, reported beside this one)")
        elif dtype == "f32" and Ph >= 4096 and Pw >= 4096 and slots <= 6 and env.get("HGS_MRAF_SPLIT", "1") != "0":
            # one column pass (col_tile_kernel RULE 3): reads GH, w, t; writes w and the two parts of the rebuilt field
            # (signal part un-normalised, noise part); the row kernel (SPLIT) reads both
            gh2 = gh * noise_cols // Pw if env.get("HGS_GH2_MASK", "1") != "0" else gh
            col = gh + 2 * P * r + w_write + gh + gh2
            row = 2 * gh + gh2
            mraf_note = ("; MRAF with a weight update in ONE column pass: the signal and the noise part of the rebuilt field "
                         f"are written separately (GH + the noise part in the {noise_cols} columns that hold a NaN target) "
                         "and joined by the row kernel")
        elif (dtype == "f64" and Pw >= 4096 and env.get("HGS_MRAF_SPLIT", "1") != "0"
              and env.get("HGS_MRAF_SPLIT64", "1") != "0"):
            # float64: one pass of the per-column kernel (reads GH, w, t; writes w, the signal part back into GH and the
            # noise part as farfield values at the NaN-target pixels), then col_kernel<LOAD | INV> over the columns that
            # hold noise (reads their farfield columns, writes their rows of the noise part); the row kernel reads both
            gh2 = gh * noise_cols // Pw
            # (float64 leaves changed weights four pixels = 32 bytes at a time: the image pixels, not whole columns)
            w_write = n_targets * r
            col = gh + 2 * P * r + w_write + gh + noise_pixels * c
            other = noise_cols * Ph * c + gh2       # the inverse-only launch (profile slot col_inv)
            row = 2 * gh + gh2
            mraf_note = ("; float64 MRAF with a weight update in ONE pass of the per-column kernel: the noise part leaves as farfield "
                         f"values ({noise_pixels} NaN-target pixels) and an inverse-only launch over the {noise_cols} columns "
                         "that hold them writes it next to GH; joined by the row kernel")
        else:
            # two column passes: forward + weight rule (reads GH, w, t; writes w), then forward + rebuild + inverse
            col = (gh + 2 * P * r + w_write) + (2 * gh + 2 * P * r)
            passes = 2
            mraf_note = "; MRAF with a weight update = two column passes"
    canon_col = (4 * P * c + (3 if wgs else 1) * P * r)
    canon_iter = ((15 if wgs else 13) * P + 2 * S) * r
    ws = gh + P * r * (2 if (wgs or mraf) else 1)           # GH + weights (+ target)
    return dict(col=col * B, col_passes=passes, row=row * B, other=other * B, other_kind=other_kind, other_name=other_name, canon_col=canon_col * B, canon_iter=canon_iter * B,
                working_set=ws * B,
                col_model=f"GH tile read + write (2 x Sh*Pw*{c} B) + weights read (P*{r}) + "
                          f"{'target read (P*%d) + ' % r if (wgs or mraf) else ''}"
                          f"{'phase_ff read (P*%d, fixed phase) + ' % r if kim else ''}"
                          f"weight writes where a weight changed ({w_write} B)"
                          + mraf_note,
                row_model=f"H read + G written, SLM rows only (2 x Sh*Pw*{c} B); the phase itself is only "
                          "written by the last row launch of a call"
                          + ("; single-pass MRAF: the noise part is read as well, in the columns where it exists" if row != 2 * gh else ""))
