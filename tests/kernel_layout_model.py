"""Lane-accurate numpy model of the index arithmetic in cuda-l2_amd/csrc/hgemm_kernel.hpp.

It replays, for one workgroup tile, exactly what the HIP kernel does with addresses:
  * LDS-DMA staging: which source chunk each lane fetches and where it lands (lane-linear),
  * the XOR-swizzled fragment reads (ds_read_b128 per lane),
  * the MFMA lane->element contracts (v_mfma_f32_16x16x32_f16 / 32x32x16_f16, operands swapped),
  * the epilogue lane->C[m][n..n+3] mapping,
and also scores every ds_read_b128 for LDS bank conflicts with the gfx950 lane groups.
The arithmetic (matrix product) is numpy; only the *addressing* is modelled.  Used by
tests/test_kernel_layout.py so that a swizzle / fragment / epilogue indexing bug is caught on CPU.
"""
from __future__ import annotations

import numpy as np

BK = 64
ROW_BYTES = 128

# ds_read_b128 is serviced in four 16-lane groups (MI355X_MICROARCH.md, LDS table)
DS_READ_B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


class Geometry:
    def __init__(self, bm, bn, wm, wn, mi, nbuf=2):
        self.BM, self.BN, self.WM, self.WN, self.MI, self.NBUF = bm, bn, wm, wn, mi, nbuf
        self.NW = wm * wn
        self.TM, self.TN = bm // wm, bn // wn
        self.FM, self.FN = self.TM // mi, self.TN // mi
        self.NI_A = bm // 8
        self.NI = (bm + bn) // 8
        self.NJ = (self.NI + self.NW - 1) // self.NW
        self.STAGE_BYTES = (bm + bn) * ROW_BYTES


def stage_tile(geo: Geometry, a_tile: np.ndarray, bt_tile: np.ndarray, k0: int, m_valid: int, n_valid: int, oob_zero: bool = False) -> np.ndarray:
    """LDS image (in halfs) of one K-step; a_tile [rows>=..][K], bt_tile likewise (tile-local rows).
    Rows past the M / N edge: the classic / s families clamp them to the last valid row (per-piece lane offsets); family q
    (round 3: one lane offset per operand + a scalar row-block step, descriptor range = the rest of the matrix) reads
    them out of range, i.e. as zeros (oob_zero).  Either way they only feed accumulators that are never stored."""
    lds = np.full(geo.STAGE_BYTES // 2, np.nan, dtype=np.float32)
    for wave in range(geo.NW):
        for j in range(geo.NJ):
            i = wave + j * geo.NW
            if i >= geo.NI:
                continue
            is_a = i < geo.NI_A
            il = i if is_a else i - geo.NI_A
            for lane in range(64):
                r = il * 8 + (lane >> 3)
                rmax = (m_valid - 1) if is_a else (n_valid - 1)
                rc = min(r, rmax)
                chunk = (lane & 7) ^ (((il & 1) << 2) | (lane >> 4))
                src = a_tile if is_a else bt_tile
                vals = src[rc, k0 + chunk * 8: k0 + chunk * 8 + 8]
                if oob_zero:
                    # family q: offset = lane part (row wave * 8 + lane / 8 of the wave's first block, chunk keyed on the
                    # WAVE's parity) + scalar j * NW * 8 rows; the two forms must name the same row and chunk
                    q = j if is_a else j - geo.NI_A // geo.NW
                    assert geo.NW % 2 == 0 and il == wave + q * geo.NW and (il & 1) == (wave & 1)
                    if r > rmax:
                        vals = np.zeros(8, dtype=src.dtype)
                dst = (i * 1024 + lane * 16) // 2
                lds[dst:dst + 8] = vals
    return lds


def frag_offsets(geo: Geometry, lane: int):
    mi = geo.MI
    ks_n = 2 if mi == 16 else 4
    lr = lane & 15 if mi == 16 else lane & 31
    lq = lane >> 4 if mi == 16 else lane >> 5
    sw = (lr >> 1) & 7
    offs = []
    for ks in range(ks_n):
        c = ks * 4 + lq if mi == 16 else ks * 2 + lq
        offs.append(lr * ROW_BYTES + ((c ^ sw) << 4))
    return offs


def mfma(mi: int, a_op: np.ndarray, b_op: np.ndarray) -> np.ndarray:
    """a_op, b_op: [64 lanes][8]; returns acc[lane][4 or 16] of D = Aop x Bop (documented gfx950 layouts)."""
    if mi == 16:
        A = np.zeros((16, 32)); B = np.zeros((32, 16))
        for lane in range(64):
            q, i = lane >> 4, lane & 15
            A[i, q * 8:q * 8 + 8] = a_op[lane]
            B[q * 8:q * 8 + 8, i] = b_op[lane]
        D = A @ B
        acc = np.zeros((64, 4))
        for lane in range(64):
            for r in range(4):
                acc[lane, r] = D[(lane >> 4) * 4 + r, lane & 15]
        return acc
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for lane in range(64):
        q, i = lane >> 5, lane & 31
        A[i, q * 8:q * 8 + 8] = a_op[lane]
        B[q * 8:q * 8 + 8, i] = b_op[lane]
    D = A @ B
    acc = np.zeros((64, 16))
    for lane in range(64):
        for r in range(16):
            acc[lane, r] = D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
    return acc


def bank_conflict_extra_cycles(byte_addrs) -> int:
    """Extra LDS cycles of one ds_read_b128 wave instruction (0 = conflict-free)."""
    extra = 0
    for group in DS_READ_B128_GROUPS:
        per_bank = {}
        for lane in group:
            a = byte_addrs[lane]
            for dw in range(4):
                bank = ((a // 4) + dw) % 64
                per_bank.setdefault(bank, set()).add((a // 4) + dw)
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


def run_tile(geo: Geometry, A: np.ndarray, Bt: np.ndarray, m0: int, n0: int, oob_zero: bool = False, k_whole: int = -1, extra_acc=None):
    """Compute the C tile at (m0, n0) the way the kernel does. Returns (C_tile dict, conflicts).
    k_whole >= 0: only the first k_whole (a multiple of 64) elements of K go through the pipeline; extra_acc: accumulators of
    the same layout added before the epilogue (the "ktail" kernel variants: whole stages + direct_k_tail)."""
    M, K = A.shape
    N = Bt.shape[0]
    if k_whole >= 0:
        K = k_whole
    assert K % BK == 0
    # tile-local views with clamped rows handled inside stage_tile
    a_tile = A[m0:]
    bt_tile = Bt[n0:]
    mi = geo.MI
    nacc = 4 if mi == 16 else 16
    acc = np.zeros((geo.NW, geo.FM, geo.FN, 64, nacc))
    conflicts = 0
    for k0 in range(0, K, BK):
        lds = stage_tile(geo, a_tile, bt_tile, k0, M - m0, N - n0, oob_zero)
        for wave in range(geo.NW):
            wave_m, wave_n = wave // geo.WN, wave % geo.WN
            a_row_base = wave_m * geo.TM * ROW_BYTES
            b_row_base = geo.BM * ROW_BYTES + wave_n * geo.TN * ROW_BYTES
            offs = [frag_offsets(geo, lane) for lane in range(64)]
            for ks in range(len(offs[0])):
                af, bf = [], []
                for i in range(geo.FM):
                    addrs = [a_row_base + i * mi * ROW_BYTES + offs[lane][ks] for lane in range(64)]
                    conflicts += bank_conflict_extra_cycles(addrs)
                    af.append(np.stack([lds[x // 2:x // 2 + 8] for x in addrs]))
                for j in range(geo.FN):
                    addrs = [b_row_base + j * mi * ROW_BYTES + offs[lane][ks] for lane in range(64)]
                    conflicts += bank_conflict_extra_cycles(addrs)
                    bf.append(np.stack([lds[x // 2:x // 2 + 8] for x in addrs]))
                for i in range(geo.FM):
                    for j in range(geo.FN):
                        acc[wave, i, j] += mfma(mi, bf[j], af[i])  # operands swapped, as in the kernel
    if extra_acc is not None:
        acc += extra_acc
    # epilogue mapping
    out = {}
    for wave in range(geo.NW):
        wave_m, wave_n = wave // geo.WN, wave % geo.WN
        for lane in range(64):
            lm = lane & 15 if mi == 16 else lane & 31
            ln = (lane >> 4) * 4 if mi == 16 else (lane >> 5) * 4
            nq = 1 if mi == 16 else 4
            for i in range(geo.FM):
                m = m0 + wave_m * geo.TM + i * mi + lm
                for j in range(geo.FN):
                    for q in range(nq):
                        n = n0 + wave_n * geo.TN + j * mi + q * 8 + ln
                        if m < M and n < N:
                            for e in range(4):
                                key = (m, n + e)
                                assert key not in out, "two lanes write the same C element"
                                out[key] = acc[wave, i, j, lane, q * 4 + e]
    return out, conflicts


def direct_k_tail(geo: Geometry, A: np.ndarray, Bt: np.ndarray, m0: int, n0: int, k0: int, k_end: int):
    """The K tail of families q and r (hgemm_kernel.hpp: direct_k_tail), restated on byte offsets: every wave loads the MFMA
    fragments of its wave tile straight from the operands through a descriptor that starts at the tile's first row and ends
    with the matrix (at most 2 GiB); lane l reads the 16 bytes at ((row block i * 16 + l & 15) * ld + k) * 2 with
    k = k0 + 32 s + 8 (l >> 4), or at that + 2^31 when k >= k_end.  An offset at or beyond the range reads zeros.
    Returns acc[wave][i][j][lane][4] (MI = 16 members only), to be added to the pipeline's accumulators."""
    assert geo.MI == 16 and k_end > k0 and (k_end - k0) % 8 == 0
    (M, lda), (N, ldb) = A.shape, Bt.shape
    flat_a, flat_b = A.reshape(-1), Bt.reshape(-1)

    def load(flat, base_elems, range_bytes, off_bytes):
        if off_bytes >= min(range_bytes, 1 << 31):
            return np.zeros(8)
        assert off_bytes % 16 == 0 and off_bytes + 16 <= range_bytes, "a real offset must stay inside the operand"
        e = base_elems + off_bytes // 2
        return flat[e:e + 8]

    acc = np.zeros((geo.NW, geo.FM, geo.FN, 64, 4))
    range_a, range_b = (M - m0) * lda * 2, (N - n0) * ldb * 2
    nslices = (k_end - k0 + 31) // 32
    for wave in range(geo.NW):
        row_a, row_b = (wave // geo.WN) * geo.TM, (wave % geo.WN) * geo.TN
        for s_ in range(nslices):
            af = [np.zeros((64, 8)) for _ in range(geo.FM)]
            bf = [np.zeros((64, 8)) for _ in range(geo.FN)]
            for lane in range(64):
                l15, lq = lane & 15, lane >> 4
                k = k0 + s_ * 32 + lq * 8
                kb = k * 2 if k < k_end else 1 << 31
                for i in range(geo.FM):
                    af[i][lane] = load(flat_a, m0 * lda, range_a, (row_a + i * 16 + l15) * lda * 2 + kb)
                for j in range(geo.FN):
                    bf[j][lane] = load(flat_b, n0 * ldb, range_b, (row_b + j * 16 + l15) * ldb * 2 + kb)
            for i in range(geo.FM):
                for j in range(geo.FN):
                    acc[wave, i, j] += mfma(16, bf[j], af[i])
    return acc


def remap_block(bid: int, nwg: int) -> int:
    """XCD-bijective block remap (kernel prologue)."""
    xcd, idx = bid % 8, bid // 8
    q, r = nwg // 8, nwg % 8
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx


def tile_of(bid: int, tiles_m: int, tiles_n: int, group_m: int):
    tiles = tiles_m * tiles_n
    split, t_id = divmod(bid, tiles)
    gsz = group_m * tiles_n
    grp = t_id // gsz
    first_m = grp * group_m
    gm = min(tiles_m - first_m, group_m)
    tin = t_id - grp * gsz
    return split, first_m + tin % gm, tin // gm


def wide_epilogue_columns(fn: int):
    """Replay of store_tile's wide path: for each fragment pair (j, j+1) and lane, the 8 consecutive
    N offsets (inside the wave tile) the lane stores after v_permlane16_swap.  permlane16_swap(vdst, src):
    vdst' rows = [v.r0, s.r0, v.r2, s.r2], src' rows = [v.r1, s.r1, v.r3, s.r3] (rows = 16-lane groups)."""
    out = {}
    for j in range(0, fn, 2):
        # per lane, the n offsets held before the swap: tile j -> a (4 values), tile j+1 -> b
        a = {lane: [16 * j + (lane >> 4) * 4 + e for e in range(4)] for lane in range(64)}
        b = {lane: [16 * (j + 1) + (lane >> 4) * 4 + e for e in range(4)] for lane in range(64)}
        for lane in range(64):
            q, l15 = lane >> 4, lane & 15
            # a' (vdst'): even rows keep a, odd rows receive b's row q-1;  b' (src'): even rows receive a's row q+1
            a_new = a[lane] if q % 2 == 0 else b[(q - 1) * 16 + l15]
            b_new = b[lane] if q % 2 == 1 else a[(q + 1) * 16 + l15]
            cols = a_new + b_new                       # store order [a0', a1', b0', b1']
            n_base = 16 * (j + (q & 1)) + 8 * (q >> 1)
            assert cols == list(range(n_base, n_base + 8)), (lane, cols, n_base)
            out[(j, lane)] = cols
    return out


# ---- persistent family (hgemm_kernel_sp.hpp): work-item walk, hybrid tail partition, slot plan ---------
def persistent_walk(bid: int, grid: int, total_items: int):
    """Mirror of persistent_walk(): logical item ids of workgroup `bid` of a `grid`-workgroup launch, in
    the order it processes them (hgemm_kernel.hpp)."""
    xcd, j = bid % 8, bid // 8
    nwg_x = grid // 8 + (1 if xcd < grid % 8 else 0)
    q, r = total_items // 8, total_items % 8
    items_x = q + (1 if xcd < r else 0)
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    count = (items_x - j + nwg_x - 1) // nwg_x if j < items_x else 0
    return [base + j + i * nwg_x for i in range(count)]


def hybrid_partition(tiles: int, ksteps: int, G: int):
    """Host side of the hybrid schedule (hgemm_api.hip): -> (full_items, tail, S, steps_per_slice) or None
    when the plain persistent launch is used.  The profitability test is not modelled (always split)."""
    if not (G > 0 and tiles > G and tiles % G):
        return None
    tail = tiles % G
    S = min(G // tail, ksteps // 4)
    if S < 2:
        return None
    per = -(-ksteps // S)
    S = -(-ksteps // per)
    return tiles - tail, tail, S, per


def streamk_min_steps(kgran: int) -> int:
    """hgemm_api.hip: streamk_min_steps -- stages closer than this to a tile boundary are not worth a cut."""
    return 2 if kgran >= 256 else 3 if kgran >= 128 else 4


def streamk_start(tiles: int, ksteps: int, G: int, min_steps: int, w: int) -> int:
    """First stage of workgroup w's run: hgemm_kernel.hpp sk_start() restated (the device closed form).
    base(w) = w * q + min(w, r) with tiles * ksteps = q * G + r; a boundary closer than min_steps to a tile boundary snaps
    onto it; a tile with fewer than 2 * min_steps stages is never cut (the boundary goes to the nearer end)."""
    total = tiles * ksteps
    if w <= 0:
        return 0
    if w >= G:
        return total
    q, r = divmod(total, G)
    x = w * q + min(w, r)
    rem = x % ksteps
    if ksteps < 2 * min_steps:
        x += -rem if rem * 2 < ksteps else ksteps - rem
    elif rem < min_steps:
        x -= rem
    elif ksteps - rem < min_steps:
        x += ksteps - rem
    return x


def streamk_owner(tiles: int, ksteps: int, G: int, min_steps: int, x: int) -> int:
    """sk_owner(): the LAST workgroup whose run starts at or before stage x (binary search, as on the device)."""
    lo, hi = 0, G
    while hi - lo > 1:
        mid = (lo + hi) >> 1
        if streamk_start(tiles, ksteps, G, min_steps, mid) <= x:
            lo = mid
        else:
            hi = mid
    return lo


def streamk_partition(tiles: int, ksteps: int, G: int, min_steps: int = 4):
    """The stream-K schedule of hgemm_tn_sk_kernel / hgemm_tn_rs_sk_kernel (hgemm_kernel.hpp: StreamK; what every hipBLASLt
    kernel on this chip runs, and the reference's H100 StreamKScheduler shapes).  The tiles x ksteps pipeline stages of a GEMM
    form one sequence, tile-major; workgroup w of G takes the contiguous run [start(w), start(w + 1)) (streamk_start).
    -> list over workgroups of segments (tile, k0, k1, slot): slot = 0 when the segment opens the workgroup's run, 1 otherwise,
    None for a segment that covers its whole tile (stored directly, no slab).  A workgroup has at most two partial segments
    (its first and its last), so 2 * G compact slabs suffice (slab id 2 * w + slot); the parts of a tile are consecutive
    workgroups in K order, the one that completes the tile's stage count adds the slabs in workgroup order (deterministic)."""
    out = []
    for w in range(G):
        a, b = streamk_start(tiles, ksteps, G, min_steps, w), streamk_start(tiles, ksteps, G, min_steps, w + 1)
        assert a <= b
        segs = []
        x = a
        while x < b:
            tile, k0 = divmod(x, ksteps)
            k1 = min(ksteps, k0 + (b - x))
            whole = k0 == 0 and k1 == ksteps
            segs.append((tile, k0, k1, None if whole else (0 if x == a else 1)))
            x += k1 - k0
        out.append(segs)
    return out


def streamk_combine_parts(tiles: int, ksteps: int, G: int, min_steps: int, tile: int):
    """sk_epilogue()'s enumeration of a cut tile's parts: -> [(slab id, k0, k1)] in K order, found from the tile id alone."""
    t0, t1 = tile * ksteps, (tile + 1) * ksteps
    parts = []
    w = streamk_owner(tiles, ksteps, G, min_steps, t0)
    while w < G:
        b, e = streamk_start(tiles, ksteps, G, min_steps, w), streamk_start(tiles, ksteps, G, min_steps, w + 1)
        if b >= t1:
            break
        lo, hi = max(b, t0), min(e, t1)
        if hi > lo:
            parts.append((2 * w + (0 if b >= t0 else 1), lo - t0, hi - t0))
        w += 1
    return parts


def tail_item(bid: int, tail_first: int, tail_tiles: int, per: int, ksteps: int):
    """Tail-pass item id -> (tile id, first K-step, number of K-steps), as map_logical() does."""
    split, t = divmod(bid, tail_tiles)
    k0 = split * per
    return tail_first + t, k0, min(ksteps, k0 + per) - k0


def sp_plan(FM: int, FN: int, NJA: int, NJB: int, RS: int = 2):
    """Slot plan of one K-step of family 's' (SpPlan): which slots carry DMA pieces and sync points."""
    T = FM * FN
    X1, Y2 = RS * FM + 4, T // 2
    a_slots = [X1 + (a * (T - 1 - X1)) // NJA for a in range(NJA)]
    bend = max((T * 33) // 64, 2 + NJB)   # B pieces go out in the first half of interval B
    b_slots = [2 + (b * (bend - 2)) // NJB for b in range(NJB)]
    return {"T": T, "X1": X1, "Y2": Y2, "a_slots": a_slots, "b_slots": b_slots,
            "NB1": sum(s < Y2 for s in b_slots)}


# ---- family "q" (hgemm_kernel_sq.hpp): early-A operand split, two sync points per pipeline stage ---------------
def sq_plan(FM: int, FN: int, PA: int, PB: int, KT: int = 1, slack64: int = 6, rs64: int = 2, MI: int = 16, slack32: int = 4,
            spread: bool = True):
    """Mirror of CfgSQ / SqPlan: slot numbers of the leading reads, the sync point, the trailing reads and the LDS-DMA
    pieces of interval A (phase 0) and interval B (phase 1).  FM, FN = MFMA tiles per wave tile (TM / MI, TN / MI);
    MI = 32 runs two k = 16 MFMA slices per K = 32 interval.
    spread (round 3, HGEMM_SQ_SPREAD): a half-tile's first D pieces go out behind the sync point that frees its region
    ("early_*"), its last E pieces in front of the NEXT interval's sync point ("late_*": interval A carries the late
    pieces of the A half-tile, interval B those of the B half-tile).  spread = False is the round-2 plan (E = 0)."""
    SL = KT * (1 if MI == 16 else 2)
    NFA, NFB, T = FM * SL, FN * SL, FM * FN * SL
    NJA, NJB = KT * PA, KT * PB
    RS = rs64 if T >= 64 else 1
    slack = slack32 if MI == 32 else slack64 if T >= 64 else (6 if T >= 32 else 2)
    P, Q = RS * NFB + slack, RS * NFA + slack
    base = {"T": T, "P": P, "Q": Q, "NJA": NJA, "NJB": NJB, "NFA": NFA, "NFB": NFB,
            "lead_A": [RS * r for r in range(NFB)], "lead_B": [RS * r for r in range(NFA)]}
    if not spread:
        STA = 2 if (T - P - 1) // (NJB + NFA) >= 2 else 1
        STB = 2 if (T - Q - 1) // (NJA + NFB) >= 2 else 1

        def interleave(first_n, second_n):     # item index of element i of the list that goes first / second
            first = [2 * i if i < second_n else second_n + i for i in range(first_n)]
            second = [2 * i + 1 if i < first_n else first_n + i for i in range(second_n)]
            return first, second

        a_piece_items, a_read_items = interleave(NJB, NFA)       # behind P: B pieces lead, A-fragment reads follow
        b_read_items, b_piece_items = interleave(NFB, NJA)       # behind Q: B-fragment reads lead, A pieces follow
        return dict(base, EA=0, EB=0, late_A=[], late_B=[],
                    pieces_A=[P + 1 + STA * i for i in a_piece_items], reads_A=[P + 1 + STA * i for i in a_read_items],
                    reads_B=[Q + 1 + STB * i for i in b_read_items], pieces_B=[Q + 1 + STB * i for i in b_piece_items])
    EA = min((NJA * P + T // 2) // T, NJA - 1)
    EB = min((NJB * Q + T // 2) // T, NJB - 1)

    def late(E, S):
        out = []
        for k in range(E):
            x = ((2 * k + 1) * S) // (2 * E)
            if RS == 2:
                x |= 1
            out.append(max(1 + k, min(x, S - 3 - (E - 1 - k))))      # M0 one slot ahead; vmcnt wait at S - 2
        return out

    def trail(N, S):
        step = 2 if S + 1 + 2 * (N - 1) < T else 1
        return [S + 1 + step * i for i in range(N)]

    def early(D, S):
        out = []
        for k in range(D):
            x = S + 1 + ((2 * k + 1) * (T - S - 1)) // (2 * D)
            if (x - S) & 1:
                x += 1
            out.append(min(x, T - 1))
        return out

    # interval A: late A pieces, sync P, trailing A-fragment reads, early B pieces; interval B: late B, Q, B reads, early A
    return dict(base, gaps=True, EA=EA, EB=EB, late_A=late(EA, P), late_B=late(EB, Q),
                reads_A=trail(NFA, P), pieces_A=early(NJB - EB, P), reads_B=trail(NFB, Q), pieces_B=early(NJA - EA, Q))


def sq_schedule_hazards(plan: dict, steps: int = 8):
    """Replay one wave's instruction stream of family q for `steps` pipeline stages and check every LDS hazard with
    the kernel's own ordering rules.  Regions: ("A", s) / ("B", s) of stage s in {0, 1}.  Events: "vm" = a counted
    s_waitcnt vmcnt(N) of the wave, "bar" = lgkmcnt(0) + s_barrier (all four waves run the same stream and leave a
    barrier together, so one stream is enough).  Rules:
      RAW  a fragment read of tile t from a region must come behind a "bar" that itself follows (or coincides with) a "vm"
           whose count covers every DMA piece of tile t into that region: the wait makes the waiting wave's own pieces
           visible, the barrier behind it everybody's.  s_waitcnt vmcnt(N) returns with at most the N YOUNGEST pieces
           outstanding (they complete in order), so a piece is guaranteed to have landed iff at least N pieces were
           issued after it BEFORE THE WAIT;
      WAR  a DMA piece into a region must come behind a "bar" that follows every read of the region's previous occupant
           (tile t-2).
    The round-2 plan and the plain spread plan wait and synchronise in one slot; the one-instruction-per-gap form
    (plan["gaps"]) waits for vmcnt two slots ahead of the barrier.  Returns a list of violations (empty = hazard-free).
    Every tile must also receive exactly its NJ pieces."""
    T, NJA, NJB = plan["T"], plan["NJA"], plan["NJB"]
    EA = plan.get("EA", 0)
    vm_lead = 2 if plan.get("gaps") else 0
    events = []   # (time, kind, arg, tile); a sync sorts before its slot's MFMA (half 0), the slot's other items behind it

    def at(interval, slot, half=0):
        return interval * 2 * (T + 2) + slot * 2 + half

    # prologue: pieces of tiles 0, 1 (A then B each), wait + barrier, reads of tile 0 (A both halves, B first half),
    # lgkmcnt(0) + barrier, early A(2)
    t0 = -10 * (T + 2)
    order = 0
    for tile in (0, 1):
        for op, n in (("A", NJA), ("B", NJB)):
            for _ in range(n):
                events.append((t0 + order, "dma", op, tile)); order += 1
    events.append((t0 + order, "vm", NJA + NJB, None)); order += 1           # wait_vmcnt<NJA + NJB>: tile 0 landed
    events.append((t0 + order, "bar", None, None)); order += 1
    for op, half in (("A", 0), ("A", 1), ("B", 0)):
        events.append((t0 + order, "read", op, (0, half))); order += 1
    events.append((t0 + order, "bar", None, None)); order += 1
    for _ in range(NJA - EA):
        events.append((t0 + order, "dma", "A", 2)); order += 1
    for t in range(steps):
        ia, ib = 2 * t, 2 * t + 1
        for s in plan["lead_A"]:
            events.append((at(ia, s, 1), "read", "B", (t, 1)))
        for s in plan.get("late_A", []):
            events.append((at(ia, s, 1), "dma", "A", t + 2))     # the A stream still points at tile t+2 here
        events.append((at(ia, plan["P"] - vm_lead, 1 if vm_lead else 0) + (1 if vm_lead else 0), "vm", NJA + NJB, None))
        events.append((at(ia, plan["P"], 0) + (0 if vm_lead else 0.5), "bar", None, None))
        for s in plan["pieces_A"]:
            events.append((at(ia, s, 1), "dma", "B", t + 2))
        for s in plan["reads_A"]:
            events.append((at(ia, s, 1), "read", "A", (t + 1, 1)))
        for s in plan["lead_B"]:
            events.append((at(ib, s, 1), "read", "A", (t + 1, 0)))
        for s in plan.get("late_B", []):
            events.append((at(ib, s, 1), "dma", "B", t + 2))
        events.append((at(ib, plan["Q"] - vm_lead, 1 if vm_lead else 0) + (1 if vm_lead else 0), "vm", NJA + NJB, None))
        events.append((at(ib, plan["Q"], 0) + (0 if vm_lead else 0.5), "bar", None, None))
        for s in plan["reads_B"]:
            events.append((at(ib, s, 1), "read", "B", (t + 1, 0)))
        for s in plan["pieces_B"]:
            events.append((at(ib, s, 1), "dma", "A", t + 3))
    events.sort(key=lambda e: e[0])
    bad = []
    dmas = [e for e in events if e[1] == "dma"]
    vms = [e for e in events if e[1] == "vm"]
    bars = [e[0] for e in events if e[1] == "bar"]
    for op, nj in (("A", NJA), ("B", NJB)):
        for tile in range(2, steps + 1):
            got = sum(1 for e in dmas if e[2] == op and e[3] == tile)
            if got != nj:
                bad.append(("PIECES", op, tile, got))
    for time, kind, op, what in events:
        if kind == "read":
            tile = what[0]
            last_piece = max(e[0] for e in dmas if e[2] == op and e[3] == tile)
            ok = False
            for wt, _, arg, _ in vms:
                if last_piece < wt < time and sum(1 for e in dmas if last_piece < e[0] < wt) >= arg and any(wt <= b < time for b in bars):
                    ok = True
                    break
            if not ok:
                bad.append(("RAW", op, what, time))
        if kind == "dma" and what >= 2:
            prev_reads = [e for e in events if e[1] == "read" and e[2] == op and e[3][0] == what - 2]
            if not prev_reads:
                bad.append(("WAR: previous occupant was never read", op, what, time))
                continue
            last_read = max(e[0] for e in prev_reads)
            if not any(last_read < b < time for b in bars):
                bad.append(("WAR", op, what, time))
    return bad


def fused_slab_offsets(threads: int, quads: int):
    """Lane-order slab of the single-launch split-K (fused_off in hgemm_kernel.hpp): float index of element e of quad x
    of thread tid.  Must be a bijection onto [0, quads * threads * 4) with 16-byte alignment per (x, tid)."""
    return {(x, tid, e): (x * threads + tid) * 4 + e for x in range(quads) for tid in range(threads) for e in range(4)}


def ragged_piece_width(addr: int, ld: int, K: int) -> int:
    """piece_width() of hgemm_registry.hip: the widest power-of-two piece (halfs) that divides the row stride, K and the
    base address alignment, so that a piece never straddles the end of a row."""
    w = 8
    while w > 1 and (ld % w or K % w or addr % (2 * w)):
        w //= 2
    return w


# ---- family "r" (hgemm_kernel_rs.hpp): LDS image [rows][BKS*2 B], chunk c of row r at slot c ^ (r & 15) ----------------
def rs_write_addrs(bks: int, threads: int = 256, p: int = 0):
    """ds_write_b128 byte addresses of chunk p of every thread of one wave-sized slice (the kernel's lds_of(p))."""
    rb, nch = bks * 2, bks * 2 // 16
    rp = threads // nch
    out = []
    for tid in range(threads):
        r0, c0 = tid // nch, tid % nch
        base = r0 * rb + ((c0 ^ (r0 & 15)) << 4)
        addr = (base ^ (128 if (rp == 8 and (p & 1)) else 0)) + p * rp * rb
        row = r0 + p * rp
        assert addr == row * rb + ((c0 ^ (row & 15)) << 4)        # the XOR-constant form equals the definition
        out.append((row, c0, addr))
    return out


def rs_write_conflicts(addrs) -> int:
    """ds_write_b128: contiguous 8-lane groups, 32 banks of 4 B (MI355X_MICROARCH.md LDS table)."""
    extra = 0
    for w in range(0, len(addrs), 64):
        for g0 in range(w, w + 64, 8):
            per_bank = {}
            for lane in range(g0, g0 + 8):
                a = addrs[lane][2]
                for dw in range(4):
                    per_bank.setdefault(((a // 4) + dw) % 32, set()).add((a // 4) + dw)
            extra += max(len(v) for v in per_bank.values()) - 1
    return extra


def rs_frag_read_addrs(bks: int, ks: int, i: int = 0):
    """ds_read_b128 addresses of one wave for fragment row block i, K slice ks: frag_lane ^ (ks << 6) + i*16*RB."""
    rb = bks * 2
    out = []
    for lane in range(64):
        l15, lq = lane & 15, lane >> 4
        frag_lane = l15 * rb + ((lq ^ l15) << 4)
        addr = (frag_lane ^ (ks << 6)) + i * 16 * rb
        row, chunk = i * 16 + l15, 4 * ks + lq
        assert addr == row * rb + ((chunk ^ (row & 15)) << 4)     # = where rs_write_addrs put (row, chunk)
        out.append(addr)
    return out


# ---- LDS-staged fp16 epilogue (sp_epilogue_staged in hgemm_kernel_sp.hpp) -----------------------------------------
def staged_epilogue_roundtrip():
    """One group (four 16x16 accumulator tiles = 16 rows x 64 columns) through a wave's 2 KiB staging buffer.
    Returns (out, write_conflicts, read_conflicts): out[(lane, h, e)] = (row, col) of the element lane `lane` holds in
    half e of the 16 bytes it reads back for row half h; conflicts = extra LDS cycles of the 4 ds_write_b64 / 2 ds_read_b128."""
    lds = {}
    wconf = 0
    for jj in range(4):
        addrs = []
        for lane in range(64):
            wrow, wq = lane & 15, lane >> 4
            a = wrow * 128 + (((jj * 2 + (wq >> 1)) ^ (wrow & 7)) << 4) + (wq & 1) * 8
            addrs.append(a)
            for e in range(4):
                assert a + 2 * e not in lds, "two lanes write the same staging bytes"
                lds[a + 2 * e] = (wrow, jj * 16 + wq * 4 + e)       # MFMA layout: row lane & 15, 4 consecutive N per lane
        for g0 in range(0, 64, 16):                                  # ds_write_b64: contiguous 16-lane groups, 32 banks
            per_bank = {}
            for lane in range(g0, g0 + 16):
                for dw in range(2):
                    per_bank.setdefault(((addrs[lane] // 4) + dw) % 32, set()).add((addrs[lane] // 4) + dw)
            wconf += max(len(v) for v in per_bank.values()) - 1
    out = {}
    rconf = 0
    for h in range(2):
        addrs = []
        for lane in range(64):
            rrow, rch = lane >> 3, lane & 7
            a = rrow * 128 + ((rch ^ (rrow & 7)) << 4) + h * 1024
            addrs.append(a)
            for e in range(8):
                out[(lane, h, e)] = lds[a + 2 * e]
        rconf += bank_conflict_extra_cycles(addrs)
    return out, wconf, rconf


# ---- family q, round 5: the K stagger of the stream cursors and the phase offset of the walk (hgemm_kernel_sq.hpp) -----------------
def sq_stream_positions(items, xcd: int, stagger: bool, stage_bytes: int = 128, lead: int = 0):
    """Mirror of SQ_LOAD_ITEM / SQ_STEP_CURSOR / SQ_ADVANCE for ONE LDS-DMA stream of a workgroup on XCD `xcd` walking `items`
    = [(k_begin_bytes, nk stages), ...]: the (item index, kbyte) of every stream position in issue order.  The kstagger variant
    enters an item `xcd * nk // 8` stages in and wraps behind its last stage; without it the walk is k_begin, k_begin + stage, ...
    `lead` extra SQ_ADVANCE calls past the last position model the stream running ahead of the MFMAs: it must stay put."""
    out = []
    if not items:
        return out
    cur = None

    def load(i):
        kb, nk = items[i]
        stag = (xcd & 7) * nk // 8 if stagger else 0
        return {"item": i, "kt": 0, "nk": nk, "kbyte0": kb, "kwrap": nk - stag, "kbyte": kb + stag * stage_bytes}

    cur = load(0)
    total = sum(nk for _, nk in items)
    for pos in range(total + lead):
        out.append((cur["item"], cur["kbyte"]))
        if cur["kt"] + 1 < cur["nk"]:                      # SQ_STEP_CURSOR
            cur["kt"] += 1
            cur["kbyte"] += stage_bytes
            if stagger and cur["kt"] == cur["kwrap"]:
                cur["kbyte"] = cur["kbyte0"]
        elif cur["item"] + 1 < len(items):                 # SQ_LOAD_ITEM(next)
            cur = load(cur["item"] + 1)
        # else: past the last step of the last item the cursor stays put (the redundant pieces re-read a valid tile)
    return out


def sq_phase_delay_cycles(bm: int, bn: int, T: int, nk0: int, j: int, groups: int, walk_count: int, mi: int = 16) -> int:
    """Mirror of the phase-offset prologue: cycles workgroup j of its XCD sleeps before its first LDS-DMA piece (rounded up to
    the 1024-cycle s_sleep granule by the loop)."""
    grp = j & (groups - 1)
    if grp == 0 or walk_count <= 1:
        return 0
    period = nk0 * 2 * T * (32 if mi == 32 else 16) + bm * bn // 12
    spacing = min(period // groups, bm * bn // 6)
    c = 0
    while c < spacing * grp:
        c += 1024
    return c
