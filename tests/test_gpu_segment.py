"""GPU segmentation (libs/encoder.cpp:507-623) against the oracle's statement of it
(oracle/svc_segment.c): region ids must be identical for every block."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(rng, mfw, mfh, n_rects, salt):
    mask = np.ones((mfh, mfw), np.uint8)
    mv = np.tile(np.array([3.0, -2.0], np.float32), (mfh, mfw, 1))
    for _ in range(n_rects):
        h, w = rng.integers(3, max(4, mfh // 3)), rng.integers(3, max(4, mfw // 3))
        y, x = rng.integers(0, mfh - h), rng.integers(0, mfw - w)
        mask[y:y + h, x:x + w] = 0
        mv[y:y + h, x:x + w] = rng.integers(-14, 15, 2)
    s = rng.random((mfh, mfw)) < salt
    mask[s] = 0
    mv[s] = rng.integers(-14, 15, (int(s.sum()), 2))
    return mask.reshape(-1), mv.reshape(-1, 2)


@pytest.mark.parametrize("mfw,mfh", [(120, 68), (80, 45), (22, 18), (240, 135)])
@pytest.mark.parametrize("conn", [4, 8])
@pytest.mark.parametrize("flags", [0, 1, 2, 3], ids=["alone", "beside", "nofork", "beside-nofork"])
def test_segment_matches_oracle(native, oracle, mfw, mfh, conn, flags):
    rng = np.random.default_rng(mfw * 7 + conn)
    frames = 5
    masks, mvs = zip(*[_scene(rng, mfw, mfh, 1 + f, 0.01 * f) for f in range(frames)])
    masks, mvs = np.stack(masks), np.stack(mvs)
    got = native.segment_frames(torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda(), mfw, mfh,
                                seed=1234, connectivity=conn, flags=flags).cpu().numpy()
    for f in range(frames):
        want = oracle.segment(masks[f], mvs[f], mfw, mfh, connectivity=conn, seed=1234 + f)
        assert np.array_equal(got[f].astype(np.uint32), want), f"frame {f}: {(got[f] != want).sum()} blocks differ"


@pytest.mark.parametrize("kw", [dict(cluster_count=1), dict(cluster_count=3, attempt_count=1, max_iter_count=2),
                                dict(morph_rect_w=5, morph_rect_h=2), dict(epsilon=50.0), dict(cluster_count=40)])
def test_segment_parameters(native, oracle, kw):
    rng = np.random.default_rng(5)
    mfw, mfh = 60, 34
    mask, mv = _scene(rng, mfw, mfh, 4, 0.03)
    got = native.segment_frames(torch.from_numpy(mask[None]).cuda(), torch.from_numpy(mv[None]).cuda(), mfw, mfh,
                                seed=9, **kw).cpu().numpy()[0]
    okw = {"attempts" if k == "attempt_count" else "max_iter" if k == "max_iter_count" else
           "morph_w" if k == "morph_rect_w" else "morph_h" if k == "morph_rect_h" else k: v for k, v in kw.items()}
    want = oracle.segment(mask, mv, mfw, mfh, seed=9, **okw)
    assert np.array_equal(got.astype(np.uint32), want)


@pytest.mark.parametrize("kw", [dict(cluster_count=100), dict(attempt_count=20), dict(cluster_count=255, attempt_count=17, max_iter_count=4)])
def test_segment_beyond_the_fused_kernels_tables(native, oracle, kw):
    """The reference's Validate admits any positive cluster / attempt count (libs/encoder.cpp:39-61).  The fused kernels hold tables for 64
    clusters and 16 attempts; beyond that -- up to the 255 / 64 of the per-call k-means -- svc_hip_segment_frames composes the per-call
    entry points frame by frame (synchronously) instead of refusing: the oracle's region ids, several frames, frame f on seed + f."""
    rng = np.random.default_rng(11)
    mfw, mfh, frames = 60, 34, 3
    masks, mvs = zip(*[_scene(rng, mfw, mfh, 3 + f, 0.04) for f in range(frames)])
    masks, mvs = np.stack(masks), np.stack(mvs)
    got = native.segment_frames(torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda(), mfw, mfh, seed=77, **kw).cpu().numpy()
    okw = {"attempts" if k == "attempt_count" else "max_iter" if k == "max_iter_count" else k: v for k, v in kw.items()}
    for f in range(frames):
        want = oracle.segment(masks[f], mvs[f], mfw, mfh, seed=77 + f, **okw)
        assert np.array_equal(got[f].astype(np.uint32), want), (kw, f)
    with pytest.raises(native.SvcError):
        native.segment_frames(torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda(), mfw, mfh, seed=77, cluster_count=256)


def test_segment_edge_cases(native, oracle):
    mfw, mfh = 30, 20
    n = mfw * mfh
    mv = np.zeros((n, 2), np.float32)
    for mask in (np.ones(n, np.uint8), np.zeros(n, np.uint8)):  # nothing / everything is foreground
        got = native.segment_frames(torch.from_numpy(mask[None]).cuda(), torch.from_numpy(mv[None]).cuda(),
                                    mfw, mfh, seed=3).cpu().numpy()[0]
        assert np.array_equal(got.astype(np.uint32), oracle.segment(mask, mv, mfw, mfh, seed=3))
    one = np.ones(n, np.uint8)
    one[5 * mfw + 7: 5 * mfw + 10] = 0  # a 1x3 sliver is removed by the 3x3 open
    one[6 * mfw + 7: 6 * mfw + 10] = 0
    got = native.segment_frames(torch.from_numpy(one[None]).cuda(), torch.from_numpy(mv[None]).cuda(), mfw, mfh).cpu().numpy()[0]
    assert np.array_equal(got.astype(np.uint32), oracle.segment(one, mv, mfw, mfh)) and not got.any()
    with pytest.raises(native.SvcError) as e:
        native.segment_frames(torch.from_numpy(one[None]).cuda(), torch.from_numpy(mv[None]).cuda(), mfw, mfh, connectivity=6)
    assert e.value.status == native.SVC_ERR_INVALID_ARG


@pytest.mark.parametrize("mfw,mfh,density", [(120, 68, 0.45), (120, 68, 0.7), (120, 68, 0.93), (120, 68, 1.0),
                                             (240, 135, 0.6), (240, 135, 0.97), (33, 31, 0.8),
                                             (480, 270, 0.35), (480, 270, 0.12)])
@pytest.mark.parametrize("flags", [0, 1, 2, 3, 4, 6, 8, 10], ids=["alone", "beside", "nofork", "beside-nofork", "wide", "wide-nofork",
                                                              "nowide", "nowide-nofork"])
def test_segment_heavy_frames(native, oracle, mfw, mfh, density, flags):
    """Scene-cut-like frames: much of the field is foreground (the 1024-lane launch; above 8 192 blocks the
    points leave the registers for LDS and, at 8K, the workspace), next to a light and an empty frame.  Above 8 192
    blocks the attempts of a heavy frame also exist as launch sequences over several workgroups (SVC_LAUNCH_WIDE = 4
    forces that form, SVC_LAUNCH_NO_WIDE = 8 the one-workgroup form; with three frames the default is the wide form)."""
    rng = np.random.default_rng(int(density * 100) + mfw)
    n = mfw * mfh
    yy, xx = np.mgrid[0:mfh, 0:mfw]
    masks, mvs = [], []
    for f in range(3):
        d = (density, 0.02, 0.0)[f]
        blob = (rng.random((mfh, mfw)) < d)
        mask = (~blob).astype(np.uint8).reshape(-1)
        mv = np.stack([np.round(6 * np.sin(xx / 17.0 + f) + rng.integers(-2, 3, (mfh, mfw))),
                       rng.integers(-9, 10, (mfh, mfw))], -1).astype(np.float32).reshape(n, 2)
        masks.append(mask); mvs.append(mv)
    masks, mvs = np.stack(masks), np.stack(mvs)
    # flags = SVC_LAUNCH_BESIDE: the shapes that fit next to bandwidth kernels (one 256-lane attempt launch for every
    # frame, lists and labelling arrays in the workspace) must give the same region ids
    got = native.segment_frames(torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda(), mfw, mfh, seed=77,
                                flags=flags).cpu().numpy()
    for f in range(3):
        want = oracle.segment(masks[f], mvs[f], mfw, mfh, seed=77 + f)
        assert np.array_equal(got[f].astype(np.uint32), want), f"frame {f}: {(got[f] != want).sum()} blocks differ"
    assert (got[0] != 0).sum() > 0.3 * n * density


@pytest.mark.parametrize("mfw,mfh", [(600, 5), (40, 30)])
def test_segment_unpacked_points_path(native, oracle, mfw, mfh):
    """Fields wider than 512 blocks, or |mv.x| >= 8192 (never produced by block matching), take the
    unpacked 64-bit path."""
    rng = np.random.default_rng(17)
    mask, mv = _scene(rng, mfw, mfh, 3, 0.02)
    if mfw <= 512:
        mv[mask == 0, 0] *= 1000.0
    got = native.segment_frames(torch.from_numpy(mask[None]).cuda(), torch.from_numpy(mv[None]).cuda(), mfw, mfh,
                                seed=21).cpu().numpy()[0]
    assert np.array_equal(got.astype(np.uint32), oracle.segment(mask, mv, mfw, mfh, seed=21))


@pytest.mark.parametrize("kw", [dict(cluster_count=1), dict(cluster_count=3, attempt_count=1, max_iter_count=2), dict(epsilon=50.0),
                                dict(cluster_count=40, attempt_count=5), dict(max_iter_count=1), dict(cluster_count=64, max_iter_count=30)])
@pytest.mark.parametrize("frames", [1, 7])
def test_segment_wide_attempts_parameters(native, oracle, kw, frames):
    """The multi-launch form of an attempt (SVC_LAUNCH_WIDE) across the k-means parameters -- one centre (no draw at all),
    one iteration (the closing launch only), an epsilon that stops after the first update, more centres than the default,
    the iteration cap reached or not -- and across workgroup counts (1 frame: 32 workgroups per attempt; 7 frames x 5
    attempts: 7): region ids == oracle/svc_segment.c == the one-workgroup form."""
    mfw, mfh = 240, 135
    n = mfw * mfh
    rng = np.random.default_rng(frames * 13 + len(kw))
    yy, xx = np.mgrid[0:mfh, 0:mfw]
    masks, mvs = [], []
    for f in range(frames):
        d = (0.9, 0.5, 0.2, 0.97, 0.05, 0.65, 0.0)[f % 7]
        mask = (~(rng.random((mfh, mfw)) < d)).astype(np.uint8).reshape(-1)
        mv = np.stack([np.round(7 * np.sin(xx / 23.0 + f) + rng.integers(-3, 4, (mfh, mfw))), rng.integers(-9, 10, (mfh, mfw))],
                      -1).astype(np.float32).reshape(n, 2)
        masks.append(mask); mvs.append(mv)
    masks, mvs = np.stack(masks), np.stack(mvs)
    tm, tv = torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda()
    wide = native.segment_frames(tm, tv, mfw, mfh, seed=5, flags=4, **kw).cpu().numpy()
    narrow = native.segment_frames(tm, tv, mfw, mfh, seed=5, flags=8, **kw).cpu().numpy()
    assert np.array_equal(wide, narrow)
    okw = {"attempts" if k == "attempt_count" else "max_iter" if k == "max_iter_count" else k: v for k, v in kw.items()}
    for f in range(min(frames, 3)):
        want = oracle.segment(masks[f], mvs[f], mfw, mfh, seed=5 + f, **okw)
        assert np.array_equal(wide[f].astype(np.uint32), want), f"frame {f}: {(wide[f] != want).sum()} blocks differ"


@pytest.mark.parametrize("flags", [4, 6], ids=["wide", "wide-nofork"])
def test_segment_wide_attempts_take_the_light_frames_too(native, oracle, flags):
    """With the multi-launch form on, EVERY frame of the batch goes through it, not only the heavy ones: frames of 1, 2, 3
    and 12 small foreground squares (9 .. 108 blocks: fewer points than the 10 clusters, fewer than one wave, fewer than
    one workgroup's chunk), of a few hundred and of 2 000 - 2 100 blocks (either side of the old light / heavy line), an
    empty one and a scene cut, in one batch: region ids == oracle/svc_segment.c == the one-workgroup form."""
    mfw, mfh = 240, 135
    n = mfw * mfh
    rng = np.random.default_rng(99)
    yy, xx = np.mgrid[0:mfh, 0:mfw]
    masks, mvs = [], []
    for squares, side in ((1, 3), (2, 3), (3, 3), (12, 3), (9, 6), (0, 0), (1, 45), (1, 46), (1, 120)):
        mask = np.ones((mfh, mfw), np.uint8)
        for q in range(squares):
            y, x = 4 + (q // 4) * (side + 3), 5 + (q % 4) * (side + 4)
            mask[y:y + side, x:x + side] = 0
        mv = np.stack([np.round(5 * np.cos(yy / 11.0) + rng.integers(-2, 3, (mfh, mfw))), rng.integers(-9, 10, (mfh, mfw))],
                      -1).astype(np.float32)
        masks.append(mask.reshape(-1)); mvs.append(mv.reshape(n, 2))
    masks, mvs = np.stack(masks), np.stack(mvs)
    tm, tv = torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda()
    wide = native.segment_frames(tm, tv, mfw, mfh, seed=31, flags=flags).cpu().numpy()
    narrow = native.segment_frames(tm, tv, mfw, mfh, seed=31, flags=8).cpu().numpy()
    assert np.array_equal(wide, narrow)
    for f in range(len(masks)):
        want = oracle.segment(masks[f], mvs[f], mfw, mfh, seed=31 + f)
        assert np.array_equal(wide[f].astype(np.uint32), want), f"frame {f}: {(wide[f] != want).sum()} blocks differ"
    assert [int((w != 0).sum()) for w in wide[:4]] == [9, 18, 27, 108] and not wide[5].any()


def test_segment_wide_attempts_with_unpacked_frames(native, oracle):
    """A frame whose points do not pack (|mv.x| >= 8192: never produced by block matching, allowed by the API) inside a batch
    that runs the multi-launch form: the head launch of the sequence gives it its whole attempt on the generic path, the other
    frames go through seeding + Lloyd launches; 240 x 135 (seeding in one launch) and 480 x 270 (one launch per centre)."""
    for mfw, mfh in ((240, 135), (480, 270)):
        n = mfw * mfh
        rng = np.random.default_rng(mfw)
        masks, mvs = [], []
        for f, d in enumerate((0.6, 0.1, 0.9)):
            mask = (~(rng.random((mfh, mfw)) < d)).astype(np.uint8).reshape(-1)
            mv = np.stack([rng.integers(-6, 7, (mfh, mfw)), rng.integers(-6, 7, (mfh, mfw))], -1).astype(np.float32).reshape(n, 2)
            if f == 1:
                mv[mask == 0, 0] *= 2000.0
            masks.append(mask); mvs.append(mv)
        masks, mvs = np.stack(masks), np.stack(mvs)
        tm, tv = torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda()
        wide = native.segment_frames(tm, tv, mfw, mfh, seed=3, flags=4).cpu().numpy()
        narrow = native.segment_frames(tm, tv, mfw, mfh, seed=3, flags=8).cpu().numpy()
        assert np.array_equal(wide, narrow)
        for f in (0, 1):
            want = oracle.segment(masks[f], mvs[f], mfw, mfh, seed=3 + f)
            assert np.array_equal(wide[f].astype(np.uint32), want), f"{mfw}x{mfh} frame {f}: {(wide[f] != want).sum()} blocks differ"


def test_segment_wide_attempts_staggered_workgroups(native, oracle):
    """SVC_LAUNCH_WIDE forced on a batch whose launch sequence has more workgroups (420 frames x 3 attempts x 2) than the
    chip holds at once (256 CUs x 8 workgroups of 4 waves): the late workgroups of a Lloyd launch start after the early
    ones have published "this attempt is over".  The verdict of a launch must never be honoured inside that same launch
    (round 3's ADVICE: WideState::done is now the step number of the launch that wrote it); attempts that converge at
    different iterations are mixed on purpose (epsilon 6: some frames stop after two updates, scene cuts run on)."""
    mfw, mfh, frames = 240, 135, 420
    n = mfw * mfh
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:mfh, 0:mfw]
    masks = np.ones((frames, n), np.uint8)
    mvs = np.zeros((frames, n, 2), np.float32)
    for f in range(frames):
        d = (0.9, 0.03, 0.3, 0.0, 0.6, 0.01)[f % 6]
        masks[f] = (~(rng.random((mfh, mfw)) < d)).astype(np.uint8).reshape(-1)
        mvs[f, :, 0] = np.round(7 * np.sin(xx / 19.0 + f) + rng.integers(-3, 4, (mfh, mfw))).reshape(-1)
        mvs[f, :, 1] = rng.integers(-9, 10, n)
    tm, tv = torch.from_numpy(masks).cuda(), torch.from_numpy(mvs).cuda()
    wide = native.segment_frames(tm, tv, mfw, mfh, seed=9, flags=4, epsilon=6.0).cpu().numpy()
    narrow = native.segment_frames(tm, tv, mfw, mfh, seed=9, flags=8, epsilon=6.0).cpu().numpy()
    assert np.array_equal(wide, narrow)
    for f in (0, 1, 2, 4, 419):
        want = oracle.segment(masks[f], mvs[f], mfw, mfh, seed=9 + f, epsilon=6.0)
        assert np.array_equal(wide[f].astype(np.uint32), want), f"frame {f}: {(wide[f] != want).sum()} blocks differ"


def test_segment_wide_rule_counts_the_sequence_length(native):
    """The automatic rule for the launch-sequence form looks at the length of the sequence as well (max_iter + 1 launches, or
    max_iter + k above 32 768 blocks): with 100 iterations asked for, the default stays with the one-workgroup form -- seen
    from outside as identical region ids whatever the rule picks, and a forced wide form beyond grid.y is refused."""
    mfw, mfh = 240, 135
    n = mfw * mfh
    rng = np.random.default_rng(5)
    mask = (~(rng.random((1, n)) < 0.8)).astype(np.uint8)
    mv = rng.integers(-7, 8, (1, n, 2)).astype(np.float32)
    tm, tv = torch.from_numpy(mask).cuda(), torch.from_numpy(mv).cuda()
    auto = native.segment_frames(tm, tv, mfw, mfh, seed=1, max_iter_count=100).cpu().numpy()
    narrow = native.segment_frames(tm, tv, mfw, mfh, seed=1, max_iter_count=100, flags=8).cpu().numpy()
    wide = native.segment_frames(tm, tv, mfw, mfh, seed=1, max_iter_count=100, flags=4).cpu().numpy()
    assert np.array_equal(auto, narrow) and np.array_equal(wide, narrow)
