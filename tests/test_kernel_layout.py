"""CPU checks of the HIP kernel's addressing (see tests/kernel_layout_model.py)."""
import re

import numpy as np
import pytest

import kernel_layout_model as klm


def _configs_from_def(pkg_dir):
    text = (pkg_dir / "csrc" / "hgemm_configs.def").read_text()
    out = []
    for m in re.finditer(r"^HGEMM_CFG\(\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", text, re.M):
        _, bm, bn, wm, wn, mi, nbuf = map(int, m.groups())
        out.append((bm, bn, wm, wn, mi, nbuf))
    return out


def test_config_table_is_consistent(pkg_dir):
    cfgs = _configs_from_def(pkg_dir)
    assert len(cfgs) >= 20
    assert len(set(cfgs)) == len(cfgs), "duplicate geometry"
    for bm, bn, wm, wn, mi, nbuf in cfgs:
        g = klm.Geometry(bm, bn, wm, wn, mi, nbuf)
        assert bm % (wm * mi) == 0 and bn % (wn * mi) == 0
        assert g.STAGE_BYTES * nbuf <= 160 * 1024
        if nbuf > 2:  # counted vmcnt assumes an even DMA piece split
            assert g.NI % g.NW == 0


@pytest.mark.parametrize("geo", [(64, 64, 2, 2, 16), (32, 32, 1, 1, 16), (64, 32, 2, 1, 16), (64, 64, 1, 1, 32), (128, 64, 2, 2, 32)])
def test_tile_matches_numpy_and_is_bank_conflict_free(geo):
    rng = np.random.default_rng(0)
    g = klm.Geometry(*geo)
    M, N, K = g.BM + 8, g.BN + 12, 128          # ragged edges: second tile row/col is partial
    A = rng.integers(-3, 4, size=(M, K)).astype(np.float32)
    Bt = rng.integers(-3, 4, size=(N, K)).astype(np.float32)   # asymmetric operands
    ref = A @ Bt.T
    for (m0, n0) in [(0, 0), (g.BM, 0), (0, g.BN), (g.BM, g.BN)]:
        # (family q's addressing -- one lane offset per operand, rows past the edge read as zeros -- needs an even wave count)
        out, conflicts = klm.run_tile(g, A, Bt, m0, n0)
        if g.NW % 2 == 0:
            out_q, _ = klm.run_tile(g, A, Bt, m0, n0, oob_zero=True)
            assert out_q == out
        assert conflicts == 0, "swizzle must make every ds_read_b128 conflict-free"
        rows = range(m0, min(M, m0 + g.BM))
        cols = range(n0, min(N, n0 + g.BN))
        assert len(out) == len(rows) * len(cols), "every in-range C element written exactly once"
        for m in rows:
            for n in cols:
                assert out[(m, n)] == ref[m, n], (m, n)


@pytest.mark.parametrize("geo", [(64, 64, 2, 2, 16), (128, 64, 2, 2, 16), (32, 64, 1, 2, 16)])
@pytest.mark.parametrize("k", [64 + 8, 128 + 24, 64 + 32, 128 + 56, 256 + 200])
def test_whole_stages_plus_direct_k_tail_match_numpy(geo, k):
    """The "ktail" kernel variants of families q and r (round 4): the whole stages of K through the pipeline, the remaining
    k % stage elements (a multiple of 8) from fragments loaded straight from the operands, lanes past K and rows past the edge
    reading zeros through the descriptor range.  Every C element of ragged edge tiles equals numpy's; the model also asserts
    that no real (unmasked) offset leaves its operand (the last row of the last tile ends with the allocation)."""
    rng = np.random.default_rng(k)
    g = klm.Geometry(*geo)
    M, N = g.BM + 8, g.BN + 12
    A = rng.integers(-3, 4, size=(M, k)).astype(np.float32)
    Bt = rng.integers(-3, 4, size=(N, k)).astype(np.float32)
    ref = A @ Bt.T
    for stage in (64, 128, 256):
        k_whole = k // stage * stage
        if k_whole == 0 or k_whole == k:
            continue
        for (m0, n0) in [(0, 0), (g.BM, g.BN), (g.BM, 0)]:
            tail = klm.direct_k_tail(g, A, Bt, m0, n0, k_whole, k)
            out, _ = klm.run_tile(g, A, Bt, m0, n0, oob_zero=True, k_whole=k_whole, extra_acc=tail)
            rows, cols = range(m0, min(M, m0 + g.BM)), range(n0, min(N, n0 + g.BN))
            assert len(out) == len(rows) * len(cols)
            for m in rows:
                for n in cols:
                    assert out[(m, n)] == ref[m, n], (stage, m0, n0, m, n)


def test_linear_lds_would_conflict():
    """Sanity of the conflict model itself: without the XOR the same reads are 4-way conflicted."""
    g = klm.Geometry(64, 64, 2, 2, 16)
    addrs = [(lane & 15) * 128 + ((lane >> 4) << 4) for lane in range(64)]
    assert klm.bank_conflict_extra_cycles(addrs) > 0
    addrs_sw = [klm.frag_offsets(g, lane)[0] for lane in range(64)]
    assert klm.bank_conflict_extra_cycles(addrs_sw) == 0


@pytest.mark.parametrize("nwg", [1, 7, 8, 9, 31, 32, 100, 256, 257, 1000])
def test_xcd_remap_is_a_bijection(nwg):
    seen = sorted(klm.remap_block(b, nwg) for b in range(nwg))
    assert seen == list(range(nwg))


@pytest.mark.parametrize("tm,tn,gm,splits", [(1, 1, 1, 1), (16, 16, 4, 1), (5, 7, 4, 1), (3, 64, 8, 2), (9, 2, 16, 3)])
def test_raster_covers_every_tile_once(tm, tn, gm, splits):
    gm = min(gm, tm)
    seen = set()
    for bid in range(tm * tn * splits):
        s, m, n = klm.tile_of(bid, tm, tn, gm)
        assert 0 <= m < tm and 0 <= n < tn and 0 <= s < splits
        seen.add((s, m, n))
    assert len(seen) == tm * tn * splits


def test_xcd_chunks_are_compact():
    """Blocks that land on one XCD (bid % 8) should cover a compact patch of the tile grid."""
    tm = tn = 16
    nwg = tm * tn
    for xcd in range(8):
        tiles = [klm.tile_of(klm.remap_block(b, nwg), tm, tn, 4)[1:] for b in range(xcd, nwg, 8)]
        rows = {t[0] for t in tiles}
        cols = {t[1] for t in tiles}
        # 32 tiles per XCD as a 4 x 8 patch -> 12 operand panels instead of up to 32
        assert len(rows) + len(cols) <= 12


@pytest.mark.parametrize("fn", [2, 4, 8])
def test_wide_epilogue_permlane_mapping(fn):
    cols = klm.wide_epilogue_columns(fn)
    for l15 in range(16):  # every C row of the fragment: each of the fn*16 columns stored exactly once
        seen = sorted(c for (j, lane), cs in cols.items() if lane & 15 == l15 for c in cs)
        assert seen == list(range(fn * 16))


@pytest.mark.parametrize("grid,total", [(256, 256), (256, 1024), (256, 784), (256, 1000), (200, 4097), (8, 3), (256, 17), (128, 129)])
def test_persistent_walk_partitions_the_items(grid, total):
    """Every work item is processed by exactly one workgroup; XCD x owns the contiguous range map_block
    would give it, and an XCD's concurrently processed items are consecutive ids (compact patch)."""
    seen = []
    for bid in range(grid):
        items = klm.persistent_walk(bid, grid, total)
        seen += items
        assert all(b > a for a, b in zip(items, items[1:]))
    assert sorted(seen) == list(range(total))
    for xcd in range(min(8, grid)):
        first_round = sorted(klm.persistent_walk(b, grid, total)[0] for b in range(xcd, grid, 8) if klm.persistent_walk(b, grid, total))
        assert first_round == list(range(first_round[0], first_round[0] + len(first_round))) if first_round else True
    if grid >= total:  # one item per workgroup: identical to the non-persistent XCD remap
        for bid in range(total):
            got = klm.persistent_walk(bid, total, total)
            assert got == [klm.remap_block(bid, total)]


@pytest.mark.parametrize("tiles,ksteps,G", [(784, 112, 256), (576, 96, 256), (289, 64, 256), (1600, 16, 256), (300, 8, 256), (380, 40, 256)])
def test_hybrid_tail_partition_covers_every_tile_and_k_step_once(tiles, ksteps, G):
    part = klm.hybrid_partition(tiles, ksteps, G)
    assert part is not None
    full, tail, S, per = part
    assert full % G == 0 and full + tail == tiles and tail * S <= G and 2 <= S
    assert (S - 1) * per < ksteps <= S * per          # no empty slice, K covered
    cover = {}
    for bid in range(tail * S):
        t, k0, nk = klm.tail_item(bid, full, tail, per, ksteps)
        assert full <= t < tiles and nk >= 1
        cover.setdefault(t, []).append((k0, nk))
    assert sorted(cover) == list(range(full, tiles))
    for t, parts in cover.items():
        parts.sort()
        assert parts[0][0] == 0 and sum(nk for _, nk in parts) == ksteps
        assert all(a[0] + a[1] == b[0] for a, b in zip(parts, parts[1:]))
    # compact slabs: slab index = tail item id, all distinct
    assert len({bid for bid in range(tail * S)}) == tail * S


def test_hybrid_schedule_is_not_used_when_it_cannot_help():
    assert klm.hybrid_partition(768, 112, 256) is None      # whole rounds only
    assert klm.hybrid_partition(200, 112, 256) is None      # a single partial round: nothing to shorten
    assert klm.hybrid_partition(300, 4, 256) is None        # K too short to slice (< 4 steps per slice)
    assert klm.hybrid_partition(511, 40, 256) is None       # tail of 255 tiles: nearly a full round already


@pytest.mark.parametrize("bm,bn", [(256, 256), (256, 128), (128, 256)])
def test_sp_slot_plan_counts_match_the_waits(bm, bn):
    """The counted vmcnt waits of family 's' assume: every DMA piece of a K-step has its own MFMA slot,
    A pieces go out after X1 in interval A, B pieces in interval B, NB1 of them ahead of Y2."""
    nw, wm, wn = 4, 2, 2
    FM, FN = bm // wm // 16, bn // wn // 16
    NJA, NJB = bm // 8 // nw, bn // 8 // nw
    p = klm.sp_plan(FM, FN, NJA, NJB)
    T = p["T"]
    assert len(set(p["a_slots"])) == NJA and len(set(p["b_slots"])) == NJB
    assert all(p["X1"] <= s < T for s in p["a_slots"]) and all(2 <= s < T for s in p["b_slots"])
    assert 2 * FM <= p["Y2"] and p["Y2"] + 2 * FN <= T and 2 * (FM + FN) <= T
    assert 0 < p["NB1"] <= NJB
    # Y1 wait: the A pieces of step+1 must have landed while B(step+1) and A(step+2) may fly: NJB + NJA younger
    # Y2 wait: B(step+1) landed while A(step+2) and the NB1 early B(step+2) pieces may fly
    assert NJA + NJB <= 63 and NJA + p["NB1"] <= 63          # vmcnt is a 6-bit counter


def _sq_members(pkg_dir=None):
    from pathlib import Path

    text = (Path(__file__).resolve().parent.parent / "cuda-l2_amd" / "csrc" / "hgemm_configs.def").read_text()
    out = [tuple(map(int, m.groups()[1:])) for m in
           re.finditer(r"^HGEMM_SQ\(\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", text, re.M)]
    assert len(out) >= 6
    return out


@pytest.mark.parametrize("bm,bn,wm,wn,kt,mi", _sq_members())
def test_family_q_schedule_has_no_lds_hazard(bm, bn, wm, wn, kt, mi):
    """Family q (hgemm_kernel_sq.hpp) orders its LDS traffic with TWO sync points per stage; the counted vmcnt and the
    slot positions are replayed here for every instantiated geometry (the knob values of the experiment builds too)."""
    nw = wm * wn
    FM, FN = bm // wm // mi, bn // wn // mi
    PA, PB = bm // 8 // nw, bn // 8 // nw
    for spread in (True, False):
        for slack, rs in [(6, 2), (2, 2), (12, 2), (6, 1)]:
            plan = klm.sq_plan(FM, FN, PA, PB, kt, slack, rs, mi, slack32=min(slack, 6), spread=spread)
            T = plan["T"]
            for pieces, reads, late, sync in ((plan["pieces_A"], plan["reads_A"], plan["late_A"], plan["P"]),
                                              (plan["pieces_B"], plan["reads_B"], plan["late_B"], plan["Q"])):
                assert len(set(pieces)) == len(pieces) and len(set(reads)) == len(reads) and len(set(late)) == len(late)
                assert max(pieces + reads) < T and min(pieces + reads) > sync
                assert all(0 <= x < sync for x in late)
                if not spread:
                    assert len(set(pieces + reads)) == len(pieces + reads) and not late
            assert len(plan["late_A"]) + len(plan["pieces_B"]) == plan["NJA"] and len(plan["late_B"]) + len(plan["pieces_A"]) == plan["NJB"]
            assert max(plan["lead_A"]) < plan["P"] and max(plan["lead_B"]) < plan["Q"]
            assert plan["NJA"] + plan["NJB"] <= 63                      # vmcnt is a 6-bit counter
            assert klm.sq_schedule_hazards(plan) == []


def test_family_q_hazard_model_catches_a_misplaced_wait():
    """The model must actually be able to fail: read the next tile's A fragments BEFORE sync P, or issue the A pieces
    before sync Q, and a hazard is reported."""
    plan = klm.sq_plan(8, 8, 8, 8)
    early_read = dict(plan, reads_A=[plan["P"] - 1 - i for i in range(len(plan["reads_A"]))])
    assert any(v[0] == "RAW" for v in klm.sq_schedule_hazards(early_read))
    early_dma = dict(plan, pieces_B=[plan["Q"] - 1 - i for i in range(len(plan["pieces_B"]))])
    assert any(v[0] == "WAR" for v in klm.sq_schedule_hazards(early_dma))
    # the spread plan's own risks: a half-tile that loses a late piece, and a late piece that lands in front of the reads of
    # the region's previous occupant (interval B's late B pieces moved into interval A, ahead of the slice-1 B reads)
    assert plan["late_A"] and plan["late_B"]
    lost = dict(plan, late_A=plan["late_A"][:-1])
    assert any(v[0] == "PIECES" for v in klm.sq_schedule_hazards(lost))
    wrong_window = dict(plan, late_B=[], pieces_A=plan["pieces_A"] + [0] * len(plan["late_B"]))
    assert any(v[0] in ("WAR", "RAW") for v in klm.sq_schedule_hazards(wrong_window))
    # one-instruction-per-gap form: the vmcnt wait sits two slots ahead of the barrier; a late piece issued behind the wait
    # is one younger piece short at the wait, which then no longer proves that the half-tile read next has landed
    assert plan["gaps"]
    behind_wait = dict(plan, late_A=plan["late_A"][:-1] + [plan["P"] - 1])
    assert any(v[0] == "RAW" for v in klm.sq_schedule_hazards(behind_wait))


def test_fused_split_k_slab_layout_is_a_bijection():
    for threads, quads in [(256, 16), (256, 64), (512, 32), (64, 4)]:
        offs = klm.fused_slab_offsets(threads, quads)
        vals = sorted(offs.values())
        assert vals == list(range(threads * quads * 4))             # BM*BN floats, each exactly once
        assert all(offs[(x, t, 0)] % 4 == 0 for x in range(quads) for t in range(threads))   # 16-byte stores
        # one store instruction of a wave = 64 consecutive threads of one quad = 1 KiB contiguous
        assert all(offs[(0, t + 1, 0)] - offs[(0, t, 0)] == 4 for t in range(63))


def test_ragged_loader_piece_width():
    assert klm.ragged_piece_width(0, 200, 200) == 8 and klm.ragged_piece_width(0, 100, 100) == 4
    assert klm.ragged_piece_width(0, 50, 50) == 2 and klm.ragged_piece_width(0, 40, 33) == 1
    assert klm.ragged_piece_width(2, 64, 64) == 1 and klm.ragged_piece_width(8, 64, 64) == 4
    for addr, ld, K in [(0, 200, 200), (4, 100, 100), (0, 66, 66), (16, 72, 40)]:
        w = klm.ragged_piece_width(addr, ld, K)
        # no piece straddles the end of a row, every piece is naturally aligned
        assert K % w == 0 and ld % w == 0 and addr % (2 * w) == 0


@pytest.mark.parametrize("bks", [128, 256])
def test_family_r_lds_image_is_consistent_and_conflict_free(bks):
    """Register-staged streaming family: every (row, chunk) written once, read back from the same address, no bank
    conflicts for the 8-lane ds_write_b128 groups nor for the 16-lane ds_read_b128 groups."""
    nch = bks * 2 // 16
    rows = 128                                   # BM + BN of the 64 x 64 tile
    seen = {}
    for p in range(rows * nch // 256):
        addrs = klm.rs_write_addrs(bks, 256, p)
        assert klm.rs_write_conflicts(addrs) == 0
        for row, c, a in addrs:
            assert (row, c) not in seen
            seen[(row, c)] = a
    assert len(seen) == rows * nch and len(set(seen.values())) == len(seen)
    for ks in range(bks // 32):
        for i in range(2):
            reads = klm.rs_frag_read_addrs(bks, ks, i)
            assert klm.bank_conflict_extra_cycles(reads) == 0
            for lane, a in enumerate(reads):
                assert seen[(i * 16 + (lane & 15), 4 * ks + (lane >> 4))] == a


def test_staged_epilogue_turns_mfma_tiles_into_full_rows():
    """sp_epilogue_staged: after the round trip through the wave's LDS buffer lane l holds, for row half h, the eight
    consecutive columns 8 * (l & 7) .. + 7 of row (l >> 3) + 8h -- i.e. one store instruction writes 8 rows x 128 B; every
    staged element is read back exactly once; the row-wise reads are conflict-free, the tile-wise writes at most 2-way."""
    out, wconf, rconf = klm.staged_epilogue_roundtrip()
    seen = set()
    for (lane, h, e), (row, col) in out.items():
        assert row == (lane >> 3) + 8 * h and col == (lane & 7) * 8 + e
        seen.add((row, col))
    assert len(seen) == 16 * 64 == len(out)
    assert rconf == 0
    assert wconf <= 4 * 4          # four ds_write_b64, four lane groups each, one extra cycle (rows r and r + 8) at most


SK_CASES = [(144, 48, 256, 4), (192, 128, 256, 3), (96, 64, 256, 3), (688, 64, 256, 4), (2304, 192, 256, 4), (18, 3, 256, 4),
            (257, 64, 256, 4), (255, 7, 256, 4), (1, 256, 256, 2), (576, 96, 512, 4), (300, 5, 256, 4), (96, 64, 512, 3),
            (128, 128, 256, 3), (97, 33, 206, 3), (4, 1000, 1024, 4), (1000, 9, 768, 4), (3, 8, 4, 4)]


def _sk_grid(tiles, ksteps, G, mn):
    """hgemm_mi355x_launch: no more workgroups than runs of min_steps stages"""
    return max(1, min(G, max(1, tiles * ksteps // mn)))


@pytest.mark.parametrize("tiles,ksteps,G,mn", SK_CASES)
def test_stream_k_partition_covers_every_stage_once_with_two_slabs_per_workgroup(tiles, ksteps, G, mn):
    """The stream-K schedule the device kernels run (kernel_layout_model.streamk_partition = hgemm_kernel.hpp sk_start): the
    invariants the slab / arrival-counter protocol relies on."""
    G = _sk_grid(tiles, ksteps, G, mn)
    part = klm.streamk_partition(tiles, ksteps, G, mn)
    assert len(part) == G
    covered = {}
    slabs = {}
    for w, segs in enumerate(part):
        partial = [sg for sg in segs if sg[3] is not None]
        assert len(partial) <= 2 and len({sg[3] for sg in partial}) == len(partial)          # one slab per slot
        assert all(sg[3] is None for sg in segs[1:-1])                                         # only the run's ends are partial
        assert [sg[0] for sg in segs] == sorted({sg[0] for sg in segs})                        # one segment per tile, in order
        for tile, k0, k1, slot in segs:
            assert 0 <= k0 < k1 <= ksteps
            for k in range(k0, k1):
                assert (tile, k) not in covered
                covered[(tile, k)] = w
            if slot is not None:
                assert 2 * w + slot not in slabs
                slabs[2 * w + slot] = (tile, k0, k1)
                assert k1 - k0 >= mn and ksteps >= 2 * mn                                      # no cut shorter than the prologue; short tiles stay whole
    assert len(covered) == tiles * ksteps
    # the parts of a tile are consecutive workgroups in K order: the combine adds their slabs in workgroup order
    for tile in range(tiles):
        owners = [covered[(tile, k)] for k in range(ksteps)]
        assert owners == sorted(owners)
    # the combiner finds exactly the producers' slabs of its tile, in K order, and their stage counts complete the tile
    cut = sorted({t for (t, _, _) in slabs.values()})
    for tile in cut:
        parts = klm.streamk_combine_parts(tiles, ksteps, G, mn, tile)
        want = sorted((k0, k1, sid) for sid, (t, k0, k1) in slabs.items() if t == tile)
        assert [(k0, k1, sid) for sid, k0, k1 in parts] == want
        assert sum(k1 - k0 for _, k0, k1 in parts) == ksteps and parts[0][1] == 0 and parts[-1][2] == ksteps
        assert all(a[2] == b[1] for a, b in zip(parts, parts[1:]))
    # balance: nobody carries more than the even share plus the snapping distance
    loads = [sum(k1 - k0 for _, k0, k1, _ in segs) for segs in part]
    share = tiles * ksteps / G
    slack = 2 * mn if ksteps >= 2 * mn else ksteps
    assert max(loads) <= share + slack + 1


@pytest.mark.parametrize("stage_bytes", [128, 256])
def test_family_q_k_stagger_visits_every_stage_once_and_both_streams_agree(stage_bytes):
    """Round 5 (hgemm_kernel_sq.hpp, EPI_KSTAGGER): the rotation lives in the stream cursor.  For every XCD and every mix of stage counts
    -- incl. fewer stages than XCDs, one-stage items, split-K chunks that start inside a row -- each item's stages are visited exactly
    once, in the rotated order x * nk / 8, ..., nk - 1, 0, ...; the A and the B stream of a workgroup produce the SAME sequence (they
    derive it from the same (XCD, stage count): round 3's knob offset them separately and was wrong for non-square members); a
    stream that runs ahead of the last item stays on a valid position; without the flag the walk is the plain one."""
    rng = np.random.default_rng(5)
    for trial in range(200):
        n_items = int(rng.integers(1, 6))
        items = [(int(rng.integers(0, 64)) * stage_bytes * 8, int(rng.integers(1, 70))) for _ in range(n_items)]
        for xcd in range(8):
            for stagger in (False, True):
                a = klm.sq_stream_positions(items, xcd, stagger, stage_bytes)
                b = klm.sq_stream_positions(items, xcd, stagger, stage_bytes, lead=3)      # the B stream / a stream running ahead
                assert b[:len(a)] == a and all(p == a[-1] for p in b[len(a):])
                pos = 0
                for i, (kb, nk) in enumerate(items):
                    seg = a[pos:pos + nk]
                    pos += nk
                    assert all(it == i for it, _ in seg)
                    want = [kb + s * stage_bytes for s in range(nk)]
                    got = [k for _, k in seg]
                    assert sorted(got) == want                                           # every stage exactly once, none outside the item
                    s0 = (xcd * nk // 8) if stagger else 0
                    assert got == want[s0:] + want[:s0]                                  # the rotation, nothing else
                assert pos == len(a)
    # the eight XCDs of a one-round launch sit nk / 8 stages apart
    starts = [klm.sq_stream_positions([(0, 256)], x, True)[0][1] // 128 for x in range(8)]
    assert starts == [0, 32, 64, 96, 128, 160, 192, 224]


def test_family_q_phase_offset_spacing():
    """The phase offset of the persistent walk (HGEMM_PLAN_PHASE_OFFSET / _OFFSET4 / both): group 0 never waits, single-item walks never
    wait, neighbouring groups are an equal share of the item period apart for a short K and an epilogue's length (BM x BN / 6 cycles)
    apart for a long one -- so a 16384 x 16384 x 4096 walk pays ~11k cycles once, not half an item (the first version cost 4-5 %)."""
    T = 64     # q256x256: FM x FN MFMA slots per interval
    for groups in (2, 4, 8):
        assert klm.sq_phase_delay_cycles(256, 256, T, 4, 0, groups, 16) == 0 and klm.sq_phase_delay_cycles(256, 256, T, 4, groups, groups, 16) == 0
        assert klm.sq_phase_delay_cycles(256, 256, T, 4, 1, groups, 1) == 0
    period_k256 = 4 * 2 * T * 16 + 256 * 256 // 12
    assert abs(klm.sq_phase_delay_cycles(256, 256, T, 4, 1, 2, 16) - period_k256 // 2) <= 1024
    assert abs(klm.sq_phase_delay_cycles(256, 256, T, 4, 3, 4, 16) - 3 * (period_k256 // 4)) <= 1024
    cap = 256 * 256 // 6
    for nk in (32, 64, 256):                                   # K = 2048 ... 16384: the cap applies as soon as a share of the period exceeds it
        for groups in (2, 4, 8):
            spacing = min((nk * 2 * T * 16 + 256 * 256 // 12) // groups, cap)
            assert spacing == cap or (nk, groups) == (32, 8)
            d = [klm.sq_phase_delay_cycles(256, 256, T, nk, j, groups, 16) for j in range(groups)]
            assert d[0] == 0 and all(abs(d[g] - g * spacing) <= 1024 for g in range(groups)), (nk, groups, d)
    assert klm.sq_phase_delay_cycles(256, 256, T, 64, 1, 2, 16) < (64 * 2 * T * 16) // 8      # far below half an item period
