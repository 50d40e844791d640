"""GPU: hop_encode_frame -- the RD spine (row a0) over the HIP kernels, through the C ABI -- against the reference encoder's own decisions (tests/golden/encoder_spine.npz):
every candidate that reaches xCheckBestMode, the per-CTU costs of cost.csv, the finished per-partition data, and the reconstruction against the spine over the CPU
restatement (which the CPU suite pins to the same goldens)."""
import os
import tempfile

import numpy as np
import pytest

from hoputil import ROOT
from hoputil import lenslet
from test_spine_cpu import FRAMES, FRAMES_MI15, FRAMES_WPP, PLAIN, key_mi15, check_against_golden, cpu_last_levels, cpu_last_rd_fraction, frame, key_of, levels_match_cbf, plain_key, run_cpu, run_cpu_plain, run_cpu_wpp, spine_cpu

pytestmark = pytest.mark.gpu


def _hp():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hophip", os.path.join(ROOT, "hevc-hop_amd", "hophip.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


SLOTS = [0, 16, 24, 48]  # candidate slots (hop_ctx_set_slots): 16 -- the SS/GT candidates of a CU are evaluated side by side; 24 -- the AMP shapes with them in the first batch; 48 -- a CU's first sub-CU with it (bench.py's setting); the results must not change


@pytest.mark.parametrize("slots", SLOTS)
@pytest.mark.parametrize("W,H,seed,sharp", FRAMES)
def test_encode_frame_equals_the_reference_encoder(W, H, seed, sharp, slots):
    hp = _hp()
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"))
    Y, Cb, Cr = frame(W, H, seed, sharp)
    ctx = hp.Context(W, H, slots=slots)
    ctx.upload_orig(Y, Cb, Cr)
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, "t.txt")
        cost, bits, dist, parts, nc = ctx.encode_frame(32, 16, 0, tp)
        text = open(tp, "rb").read()
    check_against_golden(G, key_of(W, H, seed, sharp), cost, bits, dist, parts.view(np.dtype(parts.dtype.descr)), text)
    # the reconstruction (before the loop filters) and the SS reference it was committed to
    Lc = spine_cpu()
    _, _, _, _, rec, _ = run_cpu(Lc, W, H, Y, Cb, Cr)
    for c in range(3):
        assert np.array_equal(ctx.recon_download(c), rec[c]), c
    # the levels of the chosen CUs (what encodeSlice would code): equal to the CPU spine's, and consistent with the cbf flags
    lv = ctx.levels_download()
    assert np.array_equal(lv, cpu_last_levels(Lc, len(cost))) and levels_match_cbf(lv, parts.view(np.dtype(parts.dtype.descr))) > 0
    assert np.array_equal(ctx.rd_fraction_download(), cpu_last_rd_fraction(Lc, len(cost)))
    m = 80
    assert np.array_equal(ctx.ssref_download(0)[m:m + H, m:m + W], rec[0])
    print(key_of(W, H, seed, sharp), nc, "candidates", {k: tuple(v.values()) for k, v in ctx.encode_stats().items()})
    ctx.close()


@pytest.mark.parametrize("slots", SLOTS)
@pytest.mark.parametrize("W,H,seed,lag", FRAMES_WPP)
def test_encode_frame_wavefront_equals_the_reference_with_wavefront_synchro(W, H, seed, lag, slots):
    """wpp without / with the wavefront of rows (their candidate evaluations batched into common launches) against the reference run with one substream per CTU row"""
    hp = _hp()
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"))
    Y, Cb, Cr = frame(W, H, seed, False)
    ctx = hp.Context(W, H, slots=slots)
    ctx.upload_orig(Y, Cb, Cr)
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, "t.txt")
        cost, bits, dist, parts, nc = ctx.encode_frame(32, 16, 0, tp, wpp=1, wavefront_lag=lag)
        text = open(tp, "rb").read()
    check_against_golden(G, key_of(W, H, seed, False) + "_wpp", cost, bits, dist, parts.view(np.dtype(parts.dtype.descr)), text)
    Lc = spine_cpu()
    _, _, _, _, rec, _, _ = run_cpu_wpp(Lc, W, H, Y, Cb, Cr, 0)
    for c in range(3):
        assert np.array_equal(ctx.recon_download(c), rec[c]), c
    assert np.array_equal(ctx.levels_download(), cpu_last_levels(Lc, len(cost))) and np.array_equal(ctx.rd_fraction_download(), cpu_last_rd_fraction(Lc, len(cost)))
    print(key_of(W, H, seed, False), "wpp lag", lag, nc, "candidates", {k: tuple(v.values()) for k, v in ctx.encode_stats().items()})
    ctx.close()


@pytest.mark.parametrize("bd,qp", PLAIN)
def test_encode_frame_plain_intra_configurations(bd, qp):
    """BASELINE configs 1 and 4 on the GPU: I slices at 8 bit (QP 32) and 10 bit (QP 22 / 27 / 32 / 37) against the reference run with the plain intra configurations"""
    hp = _hp()
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"))
    W, H = 136, 72
    Y, Cb, Cr = lenslet(W, H, 16, 9, bitdepth=bd)
    ctx = hp.Context(W, H, bit_depth=bd)
    ctx.upload_orig(Y, Cb, Cr)
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, "t.txt")
        cost, bits, dist, parts, nc = ctx.encode_frame(qp, 16, 0, tp, plain_intra=1)
        text = open(tp, "rb").read()
    check_against_golden(G, plain_key(bd, qp), cost, bits, dist, parts.view(np.dtype(parts.dtype.descr)), text)
    rec = run_cpu_plain(spine_cpu(), W, H, Y, Cb, Cr, qp, bd)[4]
    for c in range(3):
        assert np.array_equal(ctx.recon_download(c), rec[c]), c
    ctx.close()


@pytest.mark.parametrize("slots,groups", [(0, 1), (16, 1), (16, 2)])
def test_stacked_pictures_are_coded_as_pictures_of_their_own(slots, groups, monkeypatch):
    """three independent pictures in one stacked context (hop_ctx_set_stack), coded side by side by one hop_encode_frame: picture 0 against the reference's golden run,
    the others against contexts of their own; reconstruction and SS reference (with the margins each picture extends at ITS edges) included.  groups = 2: the pictures
    dealt to two groups with a backend, streams and worker threads each (HOP_SPINE_GROUPS), running side by side"""
    hp = _hp()
    monkeypatch.setenv("HOP_SPINE_GROUPS", str(groups))
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"))
    W, H, lag = 192, 128, 5
    pics = [frame(W, H, 7, False), frame(W, H, 8, False), frame(W, H, 11, False)]
    ctx = hp.Context(W, H, pictures=3, slots=slots)
    ctx.upload_orig(ctx.stack([p[0] for p in pics]), ctx.stack([p[1] for p in pics], True), ctx.stack([p[2] for p in pics], True))
    n = 6
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, "t.txt")
        cost, bits, dist, parts, nc = ctx.encode_frame(32, 16, 0, tp, wpp=1, wavefront_lag=lag)
        text0 = open(tp + ".0", "rb").read()
    pv = parts.view(np.dtype(parts.dtype.descr))
    check_against_golden(G, "192x128_seed7_wpp", cost[:n], bits[:n], dist[:n], pv[:n], text0)
    rec = [ctx.unstack(ctx.recon_download(c), c > 0) for c in range(3)]
    levels = ctx.levels_download(); fraction = ctx.rd_fraction_download()
    ss = ctx.ssref_download(0)
    stats = ctx.encode_stats()
    ctx.close()
    for k, (Y, Cb, Cr) in enumerate(pics):
        one = hp.Context(W, H)
        one.upload_orig(Y, Cb, Cr)
        c1, b1, d1, p1, _ = one.encode_frame(32, 16, 0, None, wpp=1, wavefront_lag=lag)
        assert np.array_equal(cost[k * n:(k + 1) * n], c1) and np.array_equal(bits[k * n:(k + 1) * n], b1) and np.array_equal(dist[k * n:(k + 1) * n], d1), k
        assert parts[k * n:(k + 1) * n].tobytes() == p1.tobytes(), k
        for c in range(3):
            assert np.array_equal(rec[c][k], one.recon_download(c)), (k, c)
        assert np.array_equal(levels[k * n:(k + 1) * n], one.levels_download()) and np.array_equal(fraction[k * n:(k + 1) * n], one.rd_fraction_download()), k
        # the picture's padded SS plane (margin 80 on every side) inside the stack's plane
        s1 = one.ssref_download(0)
        assert np.array_equal(ss[k * ctx.pitch:k * ctx.pitch + H + 160], s1), k
        one.close()
    rv = stats["rendezvous"]
    print("stack of 3:", nc, "candidates, avg batch", rv["requests"] / max(1, rv["rounds"]))
    assert rv["requests"] / max(1, rv["rounds"]) > (2.0 if groups == 1 else 1.2)



@pytest.mark.parametrize("slots", [16, 48])
@pytest.mark.parametrize("W,H,seed,lag", FRAMES_MI15)
def test_encode_frame_with_micro_image_size_15_equals_the_reference_encoder(W, H, seed, lag, slots):
    """the configuration bench.py measures (pitch-15 lenslets, --MIsize=15, candidate slots on): every candidate, cost and partition datum against the reference encoder's run
    (tests/golden/encoder_spine_mi15.npz, oracle/make_golden24.py)"""
    hp = _hp()
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine_mi15.npz"))
    Y, Cb, Cr = lenslet(W, H, 15, seed)
    ctx = hp.Context(W, H, slots=slots)                              # (16: what the picture-level binding sets, oracle/enc_shim_pic.cpp; 48: bench.py)
    ctx.upload_orig(Y, Cb, Cr)
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, "t.txt")
        if lag is None: cost, bits, dist, parts, nc = ctx.encode_frame(32, 15, 0, tp)
        else: cost, bits, dist, parts, nc = ctx.encode_frame(32, 15, 0, tp, wpp=1, wavefront_lag=lag)
        text = open(tp, "rb").read()
    check_against_golden(G, key_mi15(W, H, seed, lag), cost, bits, dist, parts.view(np.dtype(parts.dtype.descr)), text)
    ctx.close()


def test_bench_frame_top_rows_equal_the_reference_encoders_cost_csv():
    """bench.py's own picture geometry: the 7728-wide frame (hoputil.lenslet(7728, 5368, 15, 2)), its top two CTU rows coded as ONE 7728x128 picture with WaveFrontSynchro and
    candidate slots, exactly as bench.py codes the frame; all 242 per-CTU RD costs against cost.csv of the UNMODIFIED reference encoder for that picture
    (tests/golden/encoder_frame_mi15_rows2.npz).  Also exercises hop_encode_progress / hop_encode_cancel the way bench.py uses them."""
    import hashlib, json, threading, time
    hp = _hp()
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_frame_mi15_rows2.npz"))
    meta = json.loads(g["meta"].tobytes().decode())
    W, H = meta["W"], meta["H"]
    Y, Cb, Cr = lenslet(W, 5368, 15, 2)
    Y, Cb, Cr = np.ascontiguousarray(Y[:H]), np.ascontiguousarray(Cb[:H // 2]), np.ascontiguousarray(Cr[:H // 2])
    if hashlib.md5(Y.tobytes()).hexdigest() != meta["y_md5"]:
        pytest.skip("numpy's sin / cos on this host do not reproduce the golden's input frame bit for bit")
    ctx = hp.Context(W, H, slots=48)
    ctx.upload_orig(Y, Cb, Cr)
    out = {}
    th = threading.Thread(target=lambda: out.update(r=ctx.encode_frame(32, 15, 0, None, wpp=1, wavefront_lag=5)))
    th.start()
    seen = []
    while th.is_alive():
        seen.append(ctx.encode_progress()); time.sleep(0.05)
    th.join()
    cost = out["r"][0]
    assert seen == sorted(seen) and ctx.encode_progress() == len(cost) == 242
    assert np.array_equal(cost, g["cost"]), np.nonzero(cost != g["cost"])[0][:10]
    # cancelled after a few CTUs: returns early, and what it retired is the reference's
    th = threading.Thread(target=lambda: out.update(r=ctx.encode_frame(32, 15, 0, None, wpp=1, wavefront_lag=5)))
    th.start()
    while th.is_alive() and not (8 <= ctx.encode_progress() < 100): time.sleep(0.01)      # (the counter still holds the last run's 242 until the new run starts)
    ctx.encode_cancel(); th.join()
    c2 = out["r"][0]; done = c2 > 0
    assert 8 <= int(done.sum()) == ctx.encode_progress() < 242 and np.array_equal(c2[done], g["cost"][done])
    ctx.close()


def test_one_picture_ctu_rows_over_two_contexts_equal_the_reference():
    """SURVEY 8(e) on the device: hop_encode_set_shard -- ONE picture (448x192, lag-5 wavefront, candidate slots) coded by two ranks, here two contexts on the one GPU driven by
    two threads, rank g taking the CTU rows r % 2 == g; every wavefront step's finished CTUs (reconstruction block, partition data, costs, coder states) cross through the
    all-gather callback (hevc-hop_amd/shard.py:ThreadAllgather; torch.distributed's RCCL all-gather sits behind the same callback in bench.py --shard-rows).  Both ranks must
    end with the whole picture: costs, partition data and reconstruction equal the reference encoder's WaveFrontSynchro run, and the two SS references / reconstructions are
    the same samples."""
    import importlib.util, threading
    hp = _hp()
    spec = importlib.util.spec_from_file_location("hop_shard", os.path.join(ROOT, "hevc-hop_amd", "shard.py"))
    sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
    W, H, seed, lag = 448, 192, 3, 5
    Y, Cb, Cr = frame(W, H, seed, False)
    G = np.load(os.path.join(ROOT, "tests", "golden", "encoder_spine.npz"))
    key = key_of(W, H, seed, False) + "_wpp"
    ag = sh.ThreadAllgather(2, timeout=300.0)
    ctxs, out, err = [], [None, None], [None, None]
    for k in range(2):
        c = hp.Context(W, H, slots=48); c.upload_orig(Y, Cb, Cr); c.set_shard(k, 2, ag.rank(k)); ctxs.append(c)

    def run(k):
        try: out[k] = ctxs[k].encode_frame(32, 16, 0, None, wpp=1, wavefront_lag=lag)
        except BaseException as e:
            err[k] = e; ag.barrier.abort()
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert err == [None, None], err
    cols, rows = (W + 63) // 64, (H + 63) // 64
    assert [ag.rank(k).calls for k in range(2)] == [cols + lag * (rows - 1)] * 2
    assert [c.encode_progress() for c in ctxs] == [cols * len(range(k, rows, 2)) for k in range(2)]      # each rank coded exactly its rows
    recs = []
    for k in range(2):
        cost, bits, dist, parts, nc = out[k]
        assert np.array_equal(cost, G[key + "/cost"]) and np.array_equal(bits, G[key + "/bits"]) and np.array_equal(dist, G[key + "/dist"]), k
        R = G[key + "/parts"]
        for a in range(R.shape[0]):
            r, q = R[a], parts[a]
            used = r[:, 1] != 15
            for name, col in (("depth", 0), ("pred_mode", 1), ("part_size", 2), ("skip", 3), ("merge_flag", 4), ("merge_idx", 5), ("gt_flag", 6), ("tr_idx", 9)):
                assert np.array_equal(np.asarray(q[name])[used].astype(np.int16), r[used, col]), (k, a, name)
        recs.append([ctxs[k].recon_download(p) for p in range(3)])
    for p in range(3):
        assert np.array_equal(recs[0][p], recs[1][p])
    _, _, _, _, rec, _, _ = run_cpu_wpp(spine_cpu(), W, H, Y, Cb, Cr, lag)
    for p in range(3):
        assert np.array_equal(recs[0][p], rec[p])
    # switched off again: the context codes whole pictures as before
    ctxs[0].set_shard(0, 1, None)
    cost = ctxs[0].encode_frame(32, 16, 0, None, wpp=1, wavefront_lag=lag)[0]
    assert np.array_equal(cost, G[key + "/cost"])
    for c in ctxs: c.close()
