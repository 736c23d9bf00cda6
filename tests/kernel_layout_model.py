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


def stage_tile(geo: Geometry, a_tile: np.ndarray, bt_tile: np.ndarray, k0: int, m_valid: int, n_valid: int) -> np.ndarray:
    """LDS image (in halfs) of one K-step; a_tile [rows>=..][K], bt_tile likewise (tile-local rows)."""
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


def run_tile(geo: Geometry, A: np.ndarray, Bt: np.ndarray, m0: int, n0: int):
    """Compute the C tile at (m0, n0) the way the kernel does. Returns (C_tile dict, conflicts)."""
    M, K = A.shape
    N = Bt.shape[0]
    assert K % BK == 0
    # tile-local views with clamped rows handled inside stage_tile
    a_tile = A[m0:]
    bt_tile = Bt[n0:]
    mi = geo.MI
    nacc = 4 if mi == 16 else 16
    acc = np.zeros((geo.NW, geo.FM, geo.FN, 64, nacc))
    conflicts = 0
    for k0 in range(0, K, BK):
        lds = stage_tile(geo, a_tile, bt_tile, k0, M - m0, N - n0)
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


# ---- ping-pong family (hgemm_kernel_pp.hpp): K=32 half-tiles, 64-B LDS rows ------------------------
class GeometryPP:
    def __init__(self, bm, bn, wm, wn):
        self.BM, self.BN, self.WM, self.WN, self.MI = bm, bn, wm, wn, 16
        self.NW = wm * wn
        self.TM, self.TN = bm // wm, bn // wn
        self.FM, self.FN = self.TM // 16, self.TN // 16
        self.NIH_A = bm // 16
        self.NIH = (bm + bn) // 16
        self.P = self.NIH // self.NW
        self.HALF_BYTES = (bm + bn) * 64


def stage_half(geo: GeometryPP, a_tile, bt_tile, k0, m_valid, n_valid):
    lds = np.full(geo.HALF_BYTES // 2, np.nan, dtype=np.float32)
    for wave in range(geo.NW):
        for p in range(geo.P):
            piece = wave + p * geo.NW
            is_a = piece < geo.NIH_A
            il = piece if is_a else piece - geo.NIH_A
            for lane in range(64):
                r = il * 16 + (lane >> 2)
                rc = min(r, (m_valid - 1) if is_a else (n_valid - 1))
                chunk = (lane & 3) ^ ((lane >> 5) << 1)
                src = a_tile if is_a else bt_tile
                vals = src[rc, k0 + chunk * 8: k0 + chunk * 8 + 8]
                dst = (piece * 1024 + lane * 16) // 2
                lds[dst:dst + 8] = vals
    return lds


def run_tile_pp(geo: GeometryPP, A, Bt, m0, n0):
    M, K = A.shape
    N = Bt.shape[0]
    assert K % BK == 0
    a_tile, bt_tile = A[m0:], Bt[n0:]
    acc = np.zeros((geo.NW, geo.FM, geo.FN, 64, 4))
    conflicts = 0
    for k0 in range(0, K, 32):
        lds = stage_half(geo, a_tile, bt_tile, k0, M - m0, N - n0)
        for wave in range(geo.NW):
            wave_m, wave_n = wave // geo.WN, wave % geo.WN
            frag = [(lane & 15) * 64 + ((((lane >> 4) ^ (((lane & 15) >> 3) << 1))) << 4) for lane in range(64)]
            a_off = wave_m * geo.TM * 64
            b_off = geo.BM * 64 + wave_n * geo.TN * 64
            af, bf = [], []
            for i in range(geo.FM):
                addrs = [a_off + i * 16 * 64 + frag[lane] for lane in range(64)]
                conflicts += bank_conflict_extra_cycles(addrs)
                af.append(np.stack([lds[x // 2:x // 2 + 8] for x in addrs]))
            for j in range(geo.FN):
                addrs = [b_off + j * 16 * 64 + frag[lane] for lane in range(64)]
                conflicts += bank_conflict_extra_cycles(addrs)
                bf.append(np.stack([lds[x // 2:x // 2 + 8] for x in addrs]))
            for i in range(geo.FM):
                for j in range(geo.FN):
                    acc[wave, i, j] += mfma(16, bf[j], af[i])
    out = {}
    for wave in range(geo.NW):
        wave_m, wave_n = wave // geo.WN, wave % geo.WN
        for lane in range(64):
            lm, ln = lane & 15, (lane >> 4) * 4
            for i in range(geo.FM):
                m = m0 + wave_m * geo.TM + i * 16 + lm
                for j in range(geo.FN):
                    n = n0 + wave_n * geo.TN + j * 16 + ln
                    if m < M and n < N:
                        for e in range(4):
                            assert (m, n + e) not in out
                            out[(m, n + e)] = acc[wave, i, j, lane, e]
    return out, conflicts


def pp_schedule_hazards(NU: int, P: int = 4):
    """Replay the two-group schedule of hgemm_tn_pp_kernel as barrier-interval events and check
    (a) a ring slot is refilled only after both groups retired their reads of its old content,
    (b) a half-tile is read only after both groups waited for their pieces of it before a barrier.
    Returns the list of violations (empty = schedule is hazard-free by construction)."""
    bad = []
    # interval in which group g runs R(u) / M(u); a phase's events happen inside that interval
    R = lambda g, u: 2 * u + g
    Mi = lambda g, u: 2 * u + 1 + g
    issue_iv = {}   # (g, v) -> interval the group's pieces of half-tile v are issued in
    landed_iv = {}  # (g, v) -> interval at whose END the group is known to have waited for v
    for g in (0, 1):
        for v in range(min(3, NU)):
            issue_iv[(g, v)] = -1          # prologue
        for u in range(NU):
            if u + 3 < NU:
                issue_iv[(g, u + 3)] = Mi(g, u)
        landed_iv[(g, 0)] = -1             # prologue wait + barrier
        for u in range(NU):
            if u + 1 < NU:
                # G0 waits after M(u), G1 after R(u): both are interval 2u+1
                landed_iv[(g, u + 1)] = Mi(0, u) if g == 0 else R(1, u)
                # counted wait correctness: younger half-tiles issued so far by this group
                issued_upto = min(u + 3, NU - 1) if g == 0 else min(u + 2, NU - 1)
                younger = issued_upto - (u + 1)
                allowed = min(NU - 2 - u, 2) if g == 0 else min(NU - 2 - u, 1)
                if allowed > younger:
                    bad.append(("wait too weak", g, u, allowed, younger))
    for v in range(NU):
        for g in (0, 1):
            # (b) read after landing: R(v) by either group must come after both groups' waits + barrier
            for g2 in (0, 1):
                if not landed_iv[(g2, v)] < R(g, v):
                    bad.append(("read before landed", g, v, g2))
            # (a) refill of slot v%4 (half-tile v) after reads of half-tile v-4 were retired:
            # reads issued in R(g2, v-4) are retired before that group's M(v-4) MFMAs, i.e. they are
            # complete once the group has passed the barrier ending interval Mi(g2, v-4) - 1... be
            # conservative: require the refill interval to be > Mi(g2, v-4) - 1 + 0, i.e. >= Mi(g2, v-4)+1
            if v >= 4:
                for g2 in (0, 1):
                    if not issue_iv[(g, v)] >= Mi(g2, v - 4) + 1:
                        bad.append(("refill too early", g, v, g2, issue_iv[(g, v)], Mi(g2, v - 4)))
    return bad


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
