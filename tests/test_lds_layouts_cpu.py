"""Index arithmetic of the LDS images the kernels build, restated on the CPU with the kernels' own formulas and checked against the
MI355X LDS rules (guide, "LDS": a `ds_read_b128` is served in four fixed groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
28-31} and the same + 32 -- and is conflict-free when the 16 addresses of a group fall into 16 different 16-byte slots of the 256-byte
bank row).  Three layouts of round 4:
  * attn2_kernel (csrc/attention_v2.h): K rows swizzled at d = 16 / 32 / 64 / 80 / 160; the XOR sits on the DMA's SOURCE address (the LDS
    image stays lane linear) and on the fragment read -- the two must be inverse, and the read conflict-free;
  * layernorm320_kernel (csrc/norm.hip): chunk q = 64 i + lane of an 8-row block -> (row q / 40, column chunk q % 40);
  * temporal_attn_mfma_kernel (csrc/temporal.hip): Q / K / V images of frame rows padded by one 16-byte slot.
No GPU: these are statements about integers."""
import pytest

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def conflict_ways_b128(addr_of_lane):
    """Worst number of DIFFERENT addresses sharing a 16-byte slot of the bank row inside one lane group (1 = conflict-free)."""
    worst = 1
    for grp in B128_GROUPS:
        by_slot = {}
        for l in grp:
            a = addr_of_lane(l)
            by_slot.setdefault((a % 256) // 16, set()).add(a)
        worst = max(worst, max(len(v) for v in by_slot.values()))
    return worst


def conflict_free_b128(addr_of_lane):
    return conflict_ways_b128(addr_of_lane) == 1


# ------------------------------------------------------------------------------------------------ attention: K rows
def kappa(ql):
    return (ql & ~12) | ((ql & 4) << 1) | ((ql & 8) >> 1)          # bits 2 and 3 swapped (attention_v2.h: krow)


def ksw_e(D):
    s = D // 8
    return 0 if s % 2 else 1 if s % 4 else 2 if s % 8 else 3


def ksw(row, e):
    return (row >> (4 - e)) & ((1 << e) - 1) if e else 0


@pytest.mark.parametrize("D", [8, 16, 32, 40, 64, 80, 160])
def test_attention_k_swizzle_is_consistent_and_conflict_free(D):
    e, krowb = ksw_e(D), D * 2
    assert (D // 8) % (1 << e) == 0                                  # the XOR stays inside the row
    # DMA: LDS byte o (lane linear, 1 KiB per instruction) <- global (row, byte cb) of the 64-key tile
    image = {}
    for o in range(0, 64 * krowb, 16):
        row = o // krowb
        cb = (o - row * krowb) ^ (ksw(row, e) << 4)
        assert 0 <= cb < krowb and cb % 16 == 0
        image[o] = (row, cb)
    assert len(set(image.values())) == 64 * krowb // 16              # a permutation of the tile's 16-byte pieces
    # fragment read of k-step k: lane (ql, hi) wants key kappa(ql) (+ 32 sub), k-slots 16 k + 8 hi .. + 7 = byte (2 k + hi) * 16 of the row
    ks = (D + 15) // 16
    for sub in range(2):
        for k in range(ks):
            if (2 * k + 1) * 16 >= krowb and D % 16 == 8 and k == ks - 1:
                continue                                             # the folded-reference step of d = 8 / 40 reads constants for hi = 1
            def addr(lane, k=k, sub=sub):
                ql, hi = lane & 31, lane >> 5
                row = sub * 32 + kappa(ql)
                return row * krowb + (((2 * k + hi) ^ ksw(kappa(ql), e)) << 4)
            for lane in range(64):
                ql, hi = lane & 31, lane >> 5
                assert image[addr(lane)] == (sub * 32 + kappa(ql), (2 * k + hi) * 16), (D, sub, k, lane)
            if D in (40, 80, 160, 64, 32):                           # the head dims of the UNets / CLIP tower: must be conflict-free
                assert conflict_free_b128(addr), (D, sub, k)
    if D in (80, 160):                                               # ... and the unswizzled image is not (what round 4 removed)
        assert not conflict_free_b128(lambda lane: kappa(lane & 31) * krowb + (lane >> 5) * 16)


def test_attention_v_transpose_swizzle_is_conflict_free():
    """V^T tile: rows of 64 keys = 128 bytes, slot (2 k + hi) of row t*32 + ql stored at slot ^ ((ql >> 1) & 7)."""
    for t in range(5):
        for k in range(4):
            assert conflict_free_b128(lambda lane: (t * 32 + (lane & 31)) * 128 + ((((k * 2 + (lane >> 5)) ^ (((lane & 31) >> 1) & 7))) << 4))


# ------------------------------------------------------------------------------------------------ LayerNorm C = 320
def test_layernorm320_chunk_to_row_mapping():
    CCH, R, NI = 40, 8, 5
    seen = set()
    for i in range(NI):
        lo, hi = (i * 64) // CCH, (i * 64 + 63) // CCH              # the rows load index i can touch (compile-time pruning in the kernel)
        assert hi - lo <= 2
        for lane in range(64):
            q = i * 64 + lane
            r, c = q // CCH, q % CCH
            assert lo <= r <= hi and 0 <= r < R
            seen.add((r, c))
            # the five loads of a wave are ONE contiguous block: element offset of (row0 + r, chunk c) == row0 * 320 + q * 8
            assert (r * 320 + c * 8) == q * 8
    assert seen == {(r, c) for r in range(R) for c in range(CCH)}


# ------------------------------------------------------------------------------------------------ temporal attention images
@pytest.mark.parametrize("D,HG,PB,F", [(40, 8, 1, 16), (80, 4, 1, 16), (160, 2, 1, 16), (40, 4, 1, 30), (40, 8, 2, 16)])
def test_temporal_images_cover_the_rows_and_fragment_reads_are_conflict_free(D, HG, PB, F):
    cw8 = HG * D // 8                                                 # 16-byte column chunks of one pixel's row slice
    spr = PB * cw8 + 1                                                # + the padding slot
    RS, nchunk = spr * 16, F * spr
    # DMA chunk c -> (frame j, slot); every (frame, pixel, column chunk) exactly once, the padding slot re-reads slot 0
    seen = {}
    for c in range(nchunk):
        j, sl = divmod(c, spr)
        pad = sl == spr - 1
        if pad:
            sl = 0
        pl, cc = divmod(sl, cw8)
        assert 0 <= j < F and 0 <= pl < PB and 0 <= cc < cw8
        if not pad:
            assert (j, pl, cc) not in seen
            seen[(j, pl, cc)] = c * 16
    assert len(seen) == F * PB * cw8
    if PB == 1:
        assert 3 * nchunk * 16 <= 48 * 1024                          # Q + K + V images within the launcher's cap (what it picks at these sizes)
    # K / Q fragment of unit (pl, hl), k-step s: lane (m = frame, g) reads 16 bytes at frame row m, byte ubase + (32 s + 8 g) * 2
    for pl in range(PB):
        for hl in range(HG):
            ubase = (pl * HG * D + hl * D) * 2
            for s in range((D + 31) // 32):
                lanes = [l for l in range(64) if 32 * s + 8 * (l >> 4) < D]
                def addr(lane):
                    m, g = lane & 15, lane >> 4
                    return min(m, F - 1) * RS + ubase + (32 * s + 8 * g) * 2
                for lane in lanes:
                    m, g = lane & 15, lane >> 4
                    col = (ubase + (32 * s + 8 * g) * 2) // 16 - pl * cw8
                    assert seen[(min(m, F - 1), pl, col)] == addr(lane)
                if len(lanes) == 64 and F >= 16:
                    # the padding slot makes the 16 frame rows of ONE k-slice (fixed g) start in 16 different slots; a b128 lane group
                    # mixes two k-slices (g, g + 1), which leaves 2-way conflicts (SQ_LDS_BANK_CONFLICT 5.9e5 per launch at d = 40:
                    # small beside a kernel that waits on HBM; recorded here so that the number has an explanation)
                    assert conflict_ways_b128(addr) <= 2, (D, pl, hl, s)
                    for g in range(4):
                        assert len({(addr(16 * g + m) % 256) // 16 for m in range(16)}) == 16


def test_temporal_pitch_two_mod_four_would_be_conflict_free():
    """Not built: the row pitches (in 16-byte slots) that make the K / Q fragment reads conflict-free under the real lane groups are the
    ones = 2 (mod 4) -- e.g. 42 instead of 41 slots at d = 40 (one pixel x 8 heads).  Kept as the arithmetic behind the note in
    csrc/temporal.hip and profiles/r04_pmc_temporal_final.md."""
    # pitch P + 32 slots: the same residue mod 16, rows far enough apart that two lanes never name the same address (a broadcast)
    good = [P for P in range(16) if conflict_free_b128(lambda lane, P=P: (lane & 15) * (P + 32) * 16 + (lane >> 4) * 16)]
    assert good == [2, 6, 10, 14]
    assert conflict_ways_b128(lambda lane: (lane & 15) * 41 * 16 + (lane >> 4) * 16) == 2      # today's pitch


# ------------------------------------------------------------------------------------------------ streaming GEMM: prologue flavours (round 5)
def ws_swz(cpr, row):
    return (row >> 1) & 7 if cpr == 40 else row & 15               # gemm_ws.h: ws_swz<CPR>


@pytest.mark.parametrize("K", [320, 640])
def test_wsgemm_prologue_lanes_touch_exactly_the_rows_their_wave_loaded(K):
    """PRO_LNF / PRO_AFF (csrc/gemm_ws.h): a loader wave works on a landed stage after ITS OWN counted vmcnt wait only, so every 16-byte
    slot its lanes read or rewrite must have been written by a DMA instruction of the same wave; the lane map (row = 8 lw + lane / 8, chunks
    8 j + lane % 8) composed with the stage swizzle must cover each of the wave's slots exactly once (an in-place rewrite that skipped or
    doubled a slot would corrupt the tile), and chunk c of a row must be the global chunk the DMA put there."""
    cpr, tr = K // 8, 16
    stage_slots = tr * cpr
    dpt = stage_slots * 16 // 1024                                  # DMA wave-instructions per tile (1 KiB each)
    per = dpt // 2
    assert per * 64 == 8 * cpr                                       # static_assert of the kernel: a loader wave owns whole rows
    pj = cpr // 8
    # DMA: piece pidx = (lw * PER + i) * 64 + lane is LDS slot pidx (lane linear) <- global chunk (c ^ swz(r)) of row r = pidx / CPR
    holds, writer = {}, {}
    for lw in range(2):
        for i in range(per):
            for lane in range(64):
                pidx = (lw * per + i) * 64 + lane
                r, c = divmod(pidx, cpr)
                holds[pidx] = (r, c ^ ws_swz(cpr, r))
                writer[pidx] = lw
    assert sorted(holds.values()) == [(r, c) for r in range(tr) for c in range(cpr)]     # a permutation of the tile
    for lw in range(2):
        touched = []
        for lane in range(64):
            prow, psub = lw * 8 + (lane >> 3), lane & 7
            for j in range(pj):
                logical = 8 * j + psub                               # the chunk whose gamma / beta / table entries the lane keeps in registers
                slot = prow * cpr + (logical ^ ws_swz(cpr, prow))
                assert writer[slot] == lw, "a prologue lane would read a slot another wave's DMA wrote"
                assert holds[slot] == (prow, logical), "the slot does not hold the chunk the lane thinks it holds"
                touched.append(slot)
        assert sorted(touched) == list(range(lw * per * 64, (lw + 1) * per * 64))        # each of the wave's slots exactly once


def test_wsgemm_contiguous_row_blocks_cover_every_tile_once():
    """PRO_AFF streams walk contiguous row blocks (tile0 = stream * tps, tstep = 1): every tile of the matrix belongs to exactly one stream,
    whatever the remainder, and a stream of the benchmark shape stays inside one image (its table is loaded once)."""
    for M, streams, rows_per_image in [(294912, 256, 9216), (32768 + 16 * 7, 256, 2048), (36864, 256, 1024), (73728, 128, 9216)]:
        ntiles = (M + 15) // 16
        tps = (ntiles + streams - 1) // streams
        seen = []
        for s in range(streams):
            tile0 = s * tps
            seen += list(range(tile0, tile0 + max(0, min(tps, ntiles - tile0))))
        assert seen == list(range(ntiles))
        if M == 294912:
            assert all((s * tps * 16) // rows_per_image == ((s * tps + tps - 1) * 16) // rows_per_image for s in range(streams))


def test_streaming_gemm_workgroup_to_stream_mapping_uses_the_leftover_cus():
    """wsgemm_kernel's blockIdx -> (row stream, column group) map (gemm_ws.h, round 6): G groups x spx = 32 // G streams per XCD, and the
    32 - G spx CUs that leaves idle on every XCD form 8 (32 - G spx) // G extra streams (numbered XCD-major).  Every (stream, group) pair is
    owned by exactly one of the 256 workgroups, the groups of a regular stream share one XCD, an extra stream spans as few XCDs as possible."""
    for G in (1, 2, 3, 4, 5, 8, 10, 15, 16):
        spx = 32 // G
        left = 32 - G * spx
        xstreams = 8 * left // G
        streams = 8 * spx + xstreams
        owner, xcds = {}, {}
        for b in range(256):
            xcd, slot = b & 7, b >> 3
            if slot < G * spx:
                grp, stream = slot % G, xcd * spx + slot // G
            else:
                e = xcd * left + (slot - G * spx)
                if e >= xstreams * G:
                    continue
                grp, stream = e % G, 8 * spx + e // G
            assert (stream, grp) not in owner
            owner[(stream, grp)] = b
            xcds.setdefault(stream, set()).add(xcd)
        assert len(owner) == streams * G and {s for s, _ in owner} == set(range(streams))
        assert all(len(xcds[s]) == 1 for s in range(8 * spx))
        if xstreams:
            assert max(len(xcds[s]) for s in range(8 * spx, streams)) <= -(-G // left) + 1
        # G = 5 (K = N = 640): 48 + 3 streams on 255 CUs; G = 10 (GEGLU N = 2560): 24 + 1; G = 3 (N = 960): 80 + 5; G = 15 (N = 1920): 16 + 1
        assert {1: 256, 2: 128, 3: 85, 4: 64, 5: 51, 8: 32, 10: 25, 15: 17, 16: 16}[G] == streams
