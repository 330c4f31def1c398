"""CPU test of the staged grad pass's PLAN (round 5: one launch for all stages, crf_kernels.hip crf_grad_den_kernel `gd_persist`).

The reference has no such thing: its beta kernel computes a frame's gradient when it gets there (den_calculate.cu:189-227, one launch per
frame).  Here the den half of the grad pass follows the two recursions, which work towards each other, in STAGES: a 16-frame block belongs to
the first stage at which both of its rows exist, and since round 5 the stages 2.. are ONE launch whose workgroups find stage, utterance and block
from their index.  `crf_debug_stage_plan` returns what the host hands the kernel (the same functions crf_loss_fwd_bwd calls); this test walks the
grid with a Python restatement of the kernel's index arithmetic -- C integer division included -- and checks, for ragged batches and many (T, piece,
taper, frames-per-workgroup) settings, that every frame of every utterance is processed exactly once, in a workgroup that waits for a stage at
which its rows exist, and that a stage's workgroups are spread over the XCDs (workgroup i runs on XCD i % 8)."""
import numpy as np
import pytest

G = 16      # kGDFrames


def cdiv(a, b):                     # C's integer division truncates towards zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def stage_of(bound, t0, tl, lx):
    """crf_grad_den_kernel's exact test: the first stage at which the block's Q rows (forward past tl) and BP rows (backward down to t0) exist."""
    sf = sb = 1
    for k in range(1, len(bound)):
        if bound[k] < tl:
            sf = k + 1
        if bound[k] < lx - 1 - t0:
            sb = k + 1
    return max(sf, sb) if t0 < tl else 1


def walk(plan, T, lxs):
    """frames[b][t] = how often frame t of utterance b is processed; also returns per-stage XCD histograms of the workgroups with work."""
    B = len(lxs)
    bound, poff, fpb, nf = plan["bound"], plan["poff"], plan["fpb"], plan["nf"]
    ns = plan["nstage"]
    frames = [np.zeros(T, dtype=np.int32) for _ in range(B)]
    # stage 1: its own launch, every block a candidate
    for b, lx in enumerate(lxs):
        for blk in range((T + G - 1) // G):
            t0, t1 = blk * G, min(blk * G + G, T)
            tl = min(t1, lx)
            if stage_of(bound, t0, tl, lx) == 1 and t0 < tl:
                frames[b][t0:tl] += 1
    xcd = {}
    for i in range(plan["workgroups"]):
        stg = 2
        while stg + 1 <= ns and i >= poff[stg + 1]:
            stg += 1
        nfc, f = nf[stg], fpb[stg]
        assert nfc == (bound[stg] - bound[stg - 1] + G - 1) // G + 3
        nsub = G // f
        r = i - poff[stg]
        r2, b = divmod(r, B)
        blk, sub = divmod(r2, nsub)
        lx = lxs[b]
        flo = cdiv(bound[stg - 1], G) - 1
        blo = cdiv(lx - 1 - bound[stg], G) - 1
        if blk < nfc:
            blk = flo + blk
        else:
            blk = blo + (blk - nfc)
            if flo <= blk < flo + nfc:
                continue
        if blk < 0 or blk * G >= T:
            continue
        t0, t1 = blk * G, min(blk * G + G, T)
        tl = min(t1, lx)
        if stage_of(bound, t0, tl, lx) != stg:
            continue
        if f < G:
            t0 += sub * f
            t1 = min(t0 + f, t1)
            tl = min(t1, lx)
            if t0 >= tl:
                continue
        # the stage's counter says: every recursion has run bound[stg] iterations -- forward rows exist up to frame bound[stg] - 1 ... (Q_t is
        # stored by iteration t, BP_t by iteration lx - 2 - t; BP[lx-1] by the set-up)
        assert tl <= bound[stg] and lx - 1 - t0 <= bound[stg], (stg, t0, tl, lx, bound)
        frames[b][t0:tl] += 1
        xcd.setdefault(stg, np.zeros(8, dtype=np.int64))[i % 8] += 1
    return frames, xcd


@pytest.fixture(scope="module")
def C():
    from cat_amd.ctc_crf import _C
    return _C


CASES = [dict(), dict(piece=48), dict(piece=64), dict(piece=96), dict(piece=128, taper=0), dict(taper=16), dict(taper=64), dict(piece=112, taper=48),
         dict(gd_sub=8), dict(gd_sub=4, piece=64), dict(gd_sub=2, taper=16), dict(first_shift=3), dict(stages=5)]


@pytest.mark.parametrize("opts", CASES, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()) or "default")
def test_every_frame_once(C, opts):
    rng = np.random.default_rng(5)
    with C.debug_opts(**opts):
        for T in [256, 257, 300, 511, 700, 1000, 1500, 1501, 2047, 3000, 4100]:
            B = int(rng.integers(1, 9))
            lxs = [T] + [int(rng.integers(0, T + 1)) for _ in range(B - 1)]
            if T == 700:
                lxs += [0, 1, 15, 16, 17, T - 1]          # empty and tiny utterances, one frame short of the batch
            plan = C.debug_stage_plan(T, len(lxs))
            bound = plan["bound"]
            assert bound[0] == 0 and bound[-1] == T and all(a < b for a, b in zip(bound, bound[1:])), bound
            assert plan["nstage"] <= 15                   # the stage counters
            if not plan["one_launch"]:
                continue
            frames, xcd = walk(plan, T, lxs)
            for b, lx in enumerate(lxs):
                assert (frames[b][:lx] == 1).all(), (T, lx, opts, bound, np.nonzero(frames[b][:lx] != 1)[0][:8])
                assert (frames[b][lx:] == 0).all(), (T, lx, opts)


def test_default_plan_of_the_metric_shape(C):
    plan = C.debug_stage_plan(1500, 64)
    assert plan["one_launch"] and plan["piece"] == 80
    assert plan["bound"][-3:] == [1404, 1468, 1500]        # tapered: ..., 64, 32
    assert plan["bound"][1] == 752                          # nothing is complete before the recursions have met
    _, xcd = walk(plan, 1500, [1500] * 64)
    for stg, h in xcd.items():                              # work of every stage on all eight XCDs, evenly (round 5: 2.73 -> 2.92 ms when it was not)
        assert h.min() > 0 and h.max() <= 1.35 * h.mean(), (stg, h)


def test_round4_schedule_is_still_there(C):
    with C.debug_opts(gd_stage_launches=1):
        plan = C.debug_stage_plan(1500, 64)
    assert not plan["one_launch"] and plan["piece"] == 128 and plan["bound"][:3] == [0, 752, 880]
