"""GPU tests (-m gpu) of the launch-configuration machinery around the strip kernels: the ONLINE tuner (candidates are
compared on the caller's own launches -- nothing extra is launched, nothing stalls) and the no-drain destroy (one object
per image, example/steer.cpp:86).  Results never depend on any of it: every call is compared bit for bit."""
import statistics

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv():
    import cvsteer_amd
    return cvsteer_amd


def test_online_tuner_issues_no_extra_launches_and_never_stalls(cv):
    """a new 4096^2 shape: the second call costs about what a settled call costs (rounds 2-3 ran ~250 timing launches inside
    it, 25-30 ms); cvs_launch_info.tuning_launches stays 0; every call -- whatever candidate configuration it ran with --
    returns the same bits; the comparison ends by itself (cvs_launch_info.tune_state 1 -> 2) after two or three rounds of SUSTAINED turns
    (20-100 consecutive calls per candidate, cvs_tune.cpp).  The launch is the full setup (12 planes); the basis pass
    and the fused steer on a large resident image are deliberately NOT tuned (see build_candidates) -- checked at the end"""
    import torch
    from cvsteer_amd import _lib as L
    n = 4096
    img = torch.rand((n, n), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    g, h = torch.empty_like(img), torch.empty_like(img)
    f = cv.SteerableFiltersG2(None)
    assert f.launch_info()["tuning_launches"] == 0

    def call():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f.setup(img, flags=cv.SETUP_FULL)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    times, configs = [], set()
    first = None
    states = []
    for i in range(2600):
        times.append(call())
        li = f.launch_info()
        configs.add((li["block_order"], li["strip_rows"], li["state_layout"]))
        states.append(li["tune_state"])
        assert li["tuning_launches"] == 0
        if li["tune_state"] == 2 and i >= 69:
            break
        if first is None:
            first = (f.getDominantOrientationAngle().clone(), f.getDominantOrientationStrength().clone(), f.basis(3).clone())
        elif i in (1, 2, 5, 9, 14, 22, 31, 45, 69, 120, 199, 260, 333, 480, 700, 1100, 1600, 2200):
            assert torch.equal(f.getDominantOrientationAngle(), first[0]) and torch.equal(f.getDominantOrientationStrength(), first[1]) and torch.equal(f.basis(3), first[2]), i
    steady = statistics.median(times[-15:])
    assert times[1] <= 1.5 * steady + 0.15, (times[:4], steady)          # the second call is an ordinary call (+ host jitter; rounds 2-3: 25-30 ms)
    assert max(times[1:]) <= 3.0 * steady + 0.2, (max(times[1:]), steady)  # ... and so is every other one
    assert len(configs) >= 2, configs                                     # candidates did take turns on these calls
    assert states[1] == 1 and states[-1] == 2, (states[:3], states[-3:])    # compared on the caller's calls, then decided (call 0 is the handle's
                                                                             # first: a new image, its own key, nothing to compare at this size)
    assert 2 not in states[:79]                                            # ... never before two rounds of two 20-call turns
    last = f.launch_info()
    tail = set()
    for _ in range(6):
        call()
        li = f.launch_info()
        tail.add((li["block_order"], li["strip_rows"], li["state_layout"]))
    assert len(tail) == 1, tail                                           # settled: one configuration from here on
    # a second handle of the same shape starts from what the process has learnt
    f2 = cv.SteerableFiltersG2(None)
    f2.setup(img, flags=cv.SETUP_FULL)
    f2.setup(img, flags=cv.SETUP_FULL)   # (a handle's first call counts as a fresh image: its own key)
    li2 = f2.launch_info()
    assert (li2["block_order"], li2["strip_rows"], li2["state_layout"]) == next(iter(tail))
    # tuner off: the default configuration on every call
    f3 = cv.SteerableFiltersG2(None)
    f3.set_option(L.OPT_AUTOTUNE, 0)
    seen = set()
    for _ in range(8):
        f3.setup(img, flags=cv.SETUP_FULL)
        li = f3.launch_info()
        seen.add((li["block_order"], li["strip_rows"]))
    assert len(seen) <= 2 and torch.equal(f3.getDominantOrientationAngle(), first[0])   # (first call = fresh-image default, then the resident default)
    del last
    # the fused steer and the basis pass on this large resident image: one configuration from the second call on, tuner or not (no order wins
    # there, and strip heights of a few per cent are not decidable on a launch that runs at the power cap: build_candidates)
    f4 = cv.SteerableFiltersG2(None)
    seen = set()
    for k in range(30):
        if k & 1:
            f4.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h))
        else:
            f4.setup(img, flags=cv.SETUP_BASIS)
        li = f4.launch_info()
        if k >= 2:
            seen.add((li["block_order"], li["strip_rows"]))
            assert li["tune_state"] == 0
    assert len(seen) == 1, seen


def test_a_bucket_of_mixed_shapes_still_comes_to_a_decision(cv):
    """the tuner's keys are pixel-count buckets; a caller that alternates a 2048-wide and a 1600-wide image of one bucket
    (one object per image of whatever size comes along, example/steer.cpp:86) offers the XCD-column order a shape it does not fit (its column
    blocks must divide by 8) on every other call.  The comparison used to stay undecided for ever; now a candidate that cannot take its turn
    passes, is dropped after three passes, and the rounds are bounded.  Every call returns the bits of a tuner-off handle."""
    import torch
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(17)
    imgs = [torch.rand((2048, 2048), device="cuda", generator=gen), torch.rand((2700, 1600), device="cuda", generator=gen)]
    ref = cv.SteerableFiltersG2(None)
    ref.set_option(L.OPT_AUTOTUNE, 0)
    want = []
    for im in imgs:
        ref.setup(im, flags=cv.SETUP_FULL)
        want.append((ref.getDominantOrientationAngle().clone(), ref.basis(5).clone()))
    f = cv.SteerableFiltersG2(None)
    decided_at = None
    for i in range(4200):
        k = (i // 3) & 1                      # runs of three calls per shape: both shapes meet every candidate's turn
        f.setup(imgs[k], flags=cv.SETUP_FULL)
        if i % 60 == 59:                      # (the third call of a run: the image of the call before -- a call on another image is a NEW image, its own key)
            torch.cuda.synchronize()
            assert torch.equal(f.getDominantOrientationAngle(), want[k][0]) and torch.equal(f.basis(5), want[k][1]), i
            if f.launch_info()["tune_state"] == 2:
                decided_at = i
                break
    assert decided_at is not None, f.launch_info()
    seen = set()
    for i in range(12):                       # (a call on another image than the one before is a NEW image: its own key -- look at the repeats)
        k = i & 1
        f.setup(imgs[k], flags=cv.SETUP_FULL)
        f.setup(imgs[k], flags=cv.SETUP_FULL)
        li = f.launch_info()
        assert li["tune_state"] == 2
        seen.add((k, li["block_order"], li["strip_rows"]))
    assert len(seen) == 2, seen               # one configuration per shape from here on


def test_new_images_are_requested_ahead_and_results_do_not_change(cv, monkeypatch):
    """launches on NEW G2 images (another pointer than the handle's previous call) of 24 MiB and more: the waves of the launch's first
    row bands also touch the rest of the image (cvs_launch_info.warm = bands per wave; cvs_kernels_basis.hip dma_warm), every
    variant, order and kind; the same image again is not a new one.  Whatever runs, the outputs are the bits of a handle with
    CVS_OPTS warm=0, f32 and 8-bit images, ragged sizes, row ranges."""
    import ctypes as C
    import torch
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(9)
    for shape in ((2560, 2560), (1500, 3001), (2100, 3200)):
        imgs = [torch.rand(shape, device="cuda", generator=gen) for _ in range(3)]
        imgs += [(im * 255).to(torch.uint8) for im in imgs[:1]]
        big = shape[0] * shape[1] * 4 >= (24 << 20)
        monkeypatch.setenv("CVS_OPTS", "warm=0")
        plain = cv.SteerableFiltersG2(None)
        plain.set_option(L.OPT_AUTOTUNE, 0)
        refs = []
        for im in imgs:
            g, h = plain.setup_steer(im, 0.3, flags=cv.SETUP_FULL)
            assert plain.launch_info()["warm"] == 0
            refs.append([g.clone(), h.clone()] + [plain.basis(p).clone() for p in range(7)] + [plain.getDominantOrientationAngle().clone()] +
                        [o.clone() for o in plain.pipeline(im)])
        monkeypatch.delenv("CVS_OPTS")
        for order in (L.ORDER_PLAIN, L.ORDER_DYNAMIC_TAIL, L.ORDER_XCD_COLUMNS, -1):
            f = cv.SteerableFiltersG2(None)
            f.set_option(L.OPT_BLOCK_ORDER, order)
            for it in range(2 * len(imgs)):
                k = it % len(imgs)
                g, h = f.setup_steer(imgs[k], 0.3, flags=cv.SETUP_FULL)
                li = f.launch_info()
                u8 = imgs[k].dtype == torch.uint8
                assert li["warm"] == (4 if big and not u8 else 0), (shape, k, li)     # (the 8-bit images here are below 24 MiB)
                cur = [g, h] + [f.basis(p) for p in range(7)] + [f.getDominantOrientationAngle().clone()] + list(f.pipeline(imgs[k]))
                for a_, b_ in zip(cur, refs[k]):
                    assert torch.equal(a_, b_), (shape, order, it)
            f.setup(imgs[0]); f.setup(imgs[0])
            assert f.launch_info()["warm"] == 0          # the same image again: resident, nothing to request ahead
        # G4 (both half banks request ahead, round 6), the launch that emits the next pyramid level (never does), and a row range
        f4w, f4p = cv.SteerableFiltersG4(None), cv.SteerableFiltersG4(None)
        monkeypatch.setenv("CVS_OPTS", "warm=0")
        f4p.setup(imgs[1])
        assert f4p.launch_info()["warm"] == 0
        monkeypatch.delenv("CVS_OPTS")
        f4w.setup(imgs[1])
        assert f4w.launch_info()["warm"] == (4 if big else 0)
        monkeypatch.setenv("CVS_OPTS", "warm=3")
        fpw = cv.SteerableFiltersG2(None)
        lvl_w = fpw.setup_pyr(imgs[2], flags=cv.SETUP_BASIS)
        monkeypatch.delenv("CVS_OPTS")
        for p in range(11):
            assert torch.equal(f4w.basis(p), f4p.basis(p)), (shape, p)
        fpp = cv.SteerableFiltersG2(None)
        lvl_p = fpp.setup_pyr(imgs[2], flags=cv.SETUP_BASIS)
        assert fpp.launch_info()["warm"] == 0 and torch.equal(lvl_w, lvl_p) and torch.equal(fpw.basis(4), fpp.basis(4))
        fr = cv.SteerableFiltersG2(None)
        fr._like = imgs[2]
        fr._bind_stream(imgs[2])
        pl = cv.api._plane(imgs[2])
        lo, hi = shape[0] // 5, shape[0] - 7
        fr._check(cv.lib().cvs_setup_rows(fr._h, C.byref(pl), cv.SETUP_BASIS, lo, hi), "cvs_setup_rows")
        for p in (0, 6):
            assert torch.equal(fr.basis(p)[lo:hi], refs[2][2 + p][lo:hi]), (shape, p, "rows")


def test_frames_of_a_state_kept_batch_are_requested_ahead_and_results_do_not_change(cv, monkeypatch):
    """cvs_pipeline_batch with the state kept on frames of 1 Mpix and more: every frame's first row bands also request the rest of the frame
    (cvs_launch_info.warm = 2, per frame); outputs-only batches and small frames do not.  Outputs and state planes are the bits of CVS_OPTS warm=0."""
    import torch
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(13)
    for shape, want in (((1080, 1920), 2), ((300, 500), 0)):
        frames = torch.rand((5,) + shape, device="cuda", generator=gen)
        monkeypatch.setenv("CVS_OPTS", "warm=0")
        ref = cv.SteerableFiltersG2(None)
        r_out = ref.pipeline_batch(frames).clone()
        assert ref.launch_info()["warm"] == 0
        ref.select_frame(3)
        r_state = [ref.basis(p).clone() for p in range(7)] + [ref.getDominantOrientationAngle().clone()]
        monkeypatch.delenv("CVS_OPTS")
        for order in (L.ORDER_PLAIN, L.ORDER_DYNAMIC_TAIL):
            f = cv.SteerableFiltersG2(None)
            f.set_option(L.OPT_BLOCK_ORDER, order)
            out = f.pipeline_batch(frames)
            assert f.launch_info()["warm"] == want, (shape, f.launch_info())
            f.select_frame(3)
            assert torch.equal(out, r_out), (shape, order)
            for a_, b_ in zip([f.basis(p) for p in range(7)] + [f.getDominantOrientationAngle()], r_state):
                assert torch.equal(a_, b_), (shape, order)
            f.set_persist(False)
            out3 = f.pipeline_batch(frames, outputs=(5, 6, 7))
            assert f.launch_info()["warm"] == 0 and torch.equal(out3, r_out[:, 5:8])


def test_objects_come_and_go_without_draining_the_device(cv):
    """one object per image (example/steer.cpp:86 inside the parallel_for_ body), no host synchronisation anywhere in the
    loop: cvs_destroy parks the state block with an event and the next object's launch waits for it on the device.  Every
    object's outputs are checked afterwards against a long-lived handle, bit for bit; the loop must also be no slower per
    object than a loop that synchronises after every object."""
    import time
    import torch
    n = 2048
    gen = torch.Generator(device="cuda").manual_seed(11)
    imgs = [torch.rand((n, n), device="cuda", generator=gen) for _ in range(4)]
    ref_f = cv.SteerableFiltersG2(None)
    refs = []
    for im in imgs:
        g, h = ref_f.setup_steer(im, 0.3, flags=cv.SETUP_BASIS)
        refs.append((g.clone(), h.clone()))
    nobj = 48
    outs = [(torch.empty_like(imgs[0]), torch.empty_like(imgs[0])) for _ in range(nobj)]

    def loop(sync_each):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nobj):
            f = cv.SteerableFiltersG2(None)
            f.setup_steer(imgs[i & 3], 0.3, flags=cv.SETUP_BASIS, out=outs[i])
            if sync_each:
                torch.cuda.synchronize()
            del f
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nobj

    loop(False)
    t_free = min(loop(False) for _ in range(3))
    for i in range(nobj):
        assert torch.equal(outs[i][0], refs[i & 3][0]) and torch.equal(outs[i][1], refs[i & 3][1]), i
    t_sync = min(loop(True) for _ in range(3))
    assert t_free <= t_sync * 1.05, (t_free, t_sync)
    # the state of a destroyed object's successor is its own: a full setup right behind a destroy, different image
    a = cv.SteerableFiltersG2(imgs[0])
    th_a = a.getDominantOrientationAngle().clone()
    del a
    b = cv.SteerableFiltersG2(imgs[1])
    c = cv.SteerableFiltersG2(imgs[0])
    assert torch.equal(c.getDominantOrientationAngle(), th_a)
    assert not torch.equal(b.getDominantOrientationAngle(), th_a)


def test_state_layouts_give_identical_planes(cv):
    """CVS_OPT_STATE_LAYOUT: row-interleaved (default) and planar state blocks hold the same values in every plane; the
    zero-copy views differ only in their row step"""
    import torch
    from cvsteer_amd import _lib as L
    rng = np.random.default_rng(3)
    for shape in ((185, 256), (300, 1021), (1080, 1920)):
        img = torch.from_numpy(rng.random(shape, dtype=np.float32)).cuda()
        for cls, nb in ((cv.SteerableFiltersG2, 7), (cv.SteerableFiltersG4, 11)):
            hs = []
            for lay in (0, 1, 2, 3):
                f = cls(None)
                f.set_option(L.OPT_STATE_LAYOUT, lay)
                f.setup(img)
                # launch_info: 2 = all twelve G2 planes in one group -- what a full setup uses by default (option 1) or pinned (2);
                # option 3 pins the two-group form; G4 has no merged form
                assert f.launch_info()["state_layout"] == ({0: 0, 1: 2, 2: 2, 3: 1}[lay] if nb == 7 else min(lay, 1))
                hs.append(f)
            for p in range(nb):
                assert torch.equal(hs[0].basis(p), hs[2].basis(p)), (shape, p, "merged")
            if nb == 7:
                assert torch.equal(hs[0].getDominantOrientationAngle(), hs[2].getDominantOrientationAngle())
                for a_, b_ in zip(hs[0].steer(None, full=True), hs[2].steer(None, full=True)):
                    assert torch.equal(a_, b_)
                for a_, b_ in zip(hs[0].pipeline(img), hs[2].pipeline(img)):
                    assert torch.equal(a_, b_)
                x, y = shape[1] // 2, shape[0] // 3
                assert tuple(hs[0].steer_point((x, y), -0.2, full=True)) == tuple(hs[2].steer_point((x, y), -0.2, full=True))
                hs[2].setup(img, flags=cv.SETUP_BASIS)          # a basis-only launch goes back to the two-group form
                assert hs[2].launch_info()["state_layout"] == 1
            for p in range(nb):
                assert torch.equal(hs[0].basis(p), hs[1].basis(p)), (shape, p)
                assert torch.equal(hs[0].basis(p), hs[3].basis(p)), (shape, p, "two groups pinned")
            if nb == 7:
                assert torch.equal(hs[0].getDominantOrientationStrength(), hs[3].getDominantOrientationStrength())
            s0, s1 = hs[0].basis_view(1)[3], hs[1].basis_view(1)[3]
            assert s1 > s0 and s1 % s0 == 0        # planes x row length against one row length
            if nb == 7:
                assert torch.equal(hs[0].getDominantOrientationAngle(), hs[1].getDominantOrientationAngle())
                assert torch.equal(hs[0].getDominantOrientationStrength(), hs[1].getDominantOrientationStrength())
                for a_, b_ in zip(hs[0].steer(None, full=True), hs[1].steer(None, full=True)):
                    assert torch.equal(a_, b_)
                x, y = shape[1] // 3, shape[0] // 2
                assert tuple(hs[0].steer_point((x, y), 0.4, full=True)) == tuple(hs[1].steer_point((x, y), 0.4, full=True))


def test_dynamic_order_many_launches_and_graph_replay(cv):
    """block order 2000000: the tile queues are back at zero after every launch -- hundreds of launches in a row, launches of
    different shapes and variants on one handle, two handles on two streams at once, and a captured graph replayed many
    times all give the bits of the plain order"""
    import torch
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(21)
    imgs = {s: torch.rand(s, device="cuda", generator=gen) for s in ((1100, 1500), (2048, 2304), (333, 777))}
    ref = {}
    plain = cv.SteerableFiltersG2(None)
    plain.set_option(L.OPT_BLOCK_ORDER, 0)
    for s, im in imgs.items():
        g, h = plain.setup_steer(im, 0.3)
        ref[s] = (g.clone(), h.clone(), [o.clone() for o in plain.pipeline(im)])
    f = cv.SteerableFiltersG2(None)
    f.set_option(L.OPT_BLOCK_ORDER, 2000000)
    for it in range(120):
        s = list(imgs)[it % 3]
        if it % 2:
            g, h = f.setup_steer(imgs[s], 0.3)
            assert f.launch_info()["block_order"] == 2000000
            if it % 7 == 1:
                assert torch.equal(g, ref[s][0]) and torch.equal(h, ref[s][1]), it
        else:
            outs = f.pipeline(imgs[s])
            if it % 5 == 0:
                for a_, b_ in zip(outs, ref[s][2]):
                    assert torch.equal(a_, b_), it
    # two handles, two streams, launches in flight together: each handle has its own queues
    s0 = (2048, 2304)
    fa, fb = cv.SteerableFiltersG2(None), cv.SteerableFiltersG2(None)
    for x in (fa, fb):
        x.set_option(L.OPT_BLOCK_ORDER, 2000000)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a = [(torch.empty(s0, device="cuda"), torch.empty(s0, device="cuda")) for _ in range(12)]
    outs_b = [(torch.empty(s0, device="cuda"), torch.empty(s0, device="cuda")) for _ in range(12)]
    torch.cuda.synchronize()
    for i in range(12):
        with torch.cuda.stream(sa):
            fa.setup_steer(imgs[s0], 0.3, out=outs_a[i])
        with torch.cuda.stream(sb):
            fb.setup_steer(imgs[s0], 0.3, out=outs_b[i])
    torch.cuda.synchronize()
    for i in range(12):
        assert torch.equal(outs_a[i][0], ref[s0][0]) and torch.equal(outs_b[i][1], ref[s0][1]), i
    # captured and replayed
    img = imgs[s0].clone()
    g, h = torch.empty_like(img), torch.empty_like(img)
    fg = cv.SteerableFiltersG2(None)
    fg.set_option(L.OPT_BLOCK_ORDER, 2000000)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fg.setup_steer(img, 0.3, out=(g, h))
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            fg.setup_steer(img, 0.3, out=(g, h))
            fg.setup_steer(img, 0.3, out=(g, h))
    torch.cuda.synchronize()
    for rep in range(25):
        g.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g, ref[s0][0]) and torch.equal(h, ref[s0][1])
    # G4: tiles of both half banks from one set of queues
    f4p, f4d = cv.SteerableFiltersG4(None), cv.SteerableFiltersG4(None)
    f4p.set_option(L.OPT_BLOCK_ORDER, 0)
    f4d.set_option(L.OPT_BLOCK_ORDER, 2000000)
    for _ in range(3):
        a_, b_ = f4p.setup_steer(imgs[s0], 0.7), f4d.setup_steer(imgs[s0], 0.7)
        assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1])
        for p in range(11):
            assert torch.equal(f4p.basis(p), f4d.basis(p)), p


def test_dynamic_order_survives_recycled_queue_slots(cv):
    """the tile queues travel with the state block; a slot that comes back from a freed block is zeroed before it serves the
    next one (found by test_steering_under_transpose_and_mirror: a block that started on the dirty set of a recycled slot
    left its tail tiles unwritten)"""
    import torch
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(8)
    shapes = [(1080, 1920), (1200, 1600), (1536, 2048), (1100, 1500)]
    imgs = [torch.rand(s, device="cuda", generator=gen) for s in shapes]
    refs = []
    for im in imgs:
        f = cv.SteerableFiltersG4(None)
        f.set_option(L.OPT_BLOCK_ORDER, 0)
        f.setup(im)
        refs.append([f.basis(p).clone() for p in range(11)])
    for rnd in range(3):
        for k, im in enumerate(imgs):
            f = cv.SteerableFiltersG4(None)
            f.set_option(L.OPT_BLOCK_ORDER, 2000000)
            for _ in range(1 + (rnd + k) % 2):          # odd and even numbers of launches: either set may be the dirty one
                f.setup(im)
            for p in range(11):
                assert torch.equal(f.basis(p), refs[k][p]), (rnd, k, p)
            del f
            cv.lib().cvs_release_cached_memory()         # the block is freed, its slot goes back to the slab
