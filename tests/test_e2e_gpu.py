"""BASELINE.json config 5 in miniature on one GPU: TAF encode (HIP) -> YOLOX train step and eval forward."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_encode_then_train_and_eval_step():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    src = e2e.SyntheticTafSource(2, events_per_window=20_000)
    x = src.encode_batch([0, 1])
    one = src.encode_batch([0, 1], batched=False)  # per-sample launches of the general path: same bytes
    assert torch.equal(x, one)
    assert x.shape == (2, 16, 256, 320, 1, 1) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    q = (x * 255).round()
    assert torch.equal(q / 255, x)  # values are exactly the uint8 artefact / 255 (data/dataset.py:294-308)
    net = e2e.build_model(16, 2)
    tr = Trainer(net, global_batch=2, nodes=1, iters_per_epoch=10)
    l0, _ = tr.train_step(x, src.labels(2), 0)
    l1, lr = tr.train_step(x, src.labels(2), 1)
    assert np.isfinite(l0) and np.isfinite(l1) and lr > 0
    net.eval()
    with torch.no_grad():
        dets = net(x)
    assert len(dets) == 2 and dets[0].shape[1] == 6


def test_batched_encode_equals_per_sample():
    """11 sequences through one frlw_taf_encode_batch call (sample 3 starts two windows late, sample 7 has no events in
    its last windows): same bytes as 11 separate general-path encodes -- every sequence follows its own
    window-without-events rule (generate_taf.py:40-41)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import e2e
    src = e2e.SyntheticTafSource(11, seed=321, events_per_window=6_000)
    # thin two samples out on the device copy: record = (t: u32, word: u32)
    rec = src.dat.view(torch.int32).reshape(-1, 2)
    t = rec[:, 0]
    lo3, hi3, lo7, hi7 = (int(src.offsets[i]) for i in (3, 4, 7, 8))
    keep = torch.ones(len(t), dtype=torch.bool, device=t.device)
    keep[lo3:hi3] = t[lo3:hi3] >= 20_000
    keep[lo7:hi7] = t[lo7:hi7] < 50_000
    counts = [int(keep[int(src.offsets[i]):int(src.offsets[i + 1])].sum()) for i in range(11)]
    src.dat = src.dat[keep].contiguous()
    src.offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    x = src.encode_batch(list(range(11)))
    one = src.encode_batch(list(range(11)), batched=False)
    assert x.shape == (11, 16, 256, 320, 1, 1)
    assert torch.equal(x, one)
    assert not torch.equal(x[3], x[4])


def test_train_batch64_is_finite_and_deterministic():
    """BASELINE.json configs[4] at its stated per-GPU size on one GPU: 64 GEN1 streams -> TAF -> one train step, twice
    from the same initial weights: finite, and the same loss to the last bit (the whole step is deterministic)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    src = e2e.SyntheticTafSource(64, seed=77, events_per_window=125_000)  # the config's stated size
    x = src.encode_batch(list(range(64)))
    assert torch.equal(x[:4], src.encode_batch([0, 1, 2, 3], batched=False))
    losses = []
    for _ in range(2):
        torch.manual_seed(0)
        tr = Trainer(e2e.build_model(16, 2), global_batch=64, nodes=1, iters_per_epoch=10)
        losses.append(tr.train_step(x, src.labels(64), 0)[0])
    assert np.isfinite(losses[0]) and losses[0] == losses[1]


def test_entry_points_train_then_test(tmp_path):
    """``train.py`` and ``test.py`` as the README launches them (one process, env:// rendezvous), on synthetic streams:
    one epoch (train batches, last_epoch checkpoint, validation through the gfx950 engine -> best_epoch), then the
    evaluation entry point loads that checkpoint and records summarise.npz."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FRLW_MAX_EPOCHS="1", FRLW_SYNTHETIC_BATCHES="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    log = str(tmp_path) + "/"
    r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--dataset", "gen1", "--batch_size", "4",
                        "--augmentation", "True", "--exp_name", "E", "--exp_type", "yolox", "--event_volume_bins", "8",
                        "--nodes", "1", "--log_path", log], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ck = os.path.join(log, "E", "checkpoints")
    assert sorted(os.listdir(ck)) == ["best_epoch.pth", "best_epoch_backbone.pth", "best_epoch_neck.pth", "last_epoch.pth",
                                      "last_epoch_backbone.pth", "last_epoch_neck.pth"]
    r = subprocess.run([sys.executable, os.path.join(root, "test.py"), "--dataset", "gen1", "--batch_size", "2", "--record", "True",
                        "--resume_exp", "E", "--exp_type", "yolox", "--event_volume_bins", "8", "--nodes", "1",
                        "--log_path", log], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "'images':" in r.stdout and os.path.exists(os.path.join(log, "E", "summarise.npz"))


def test_encode_ahead_feeds_the_same_batches():
    """e2e.EncodeAhead (the next batch encoded on a second stream while a step runs): the steps see exactly the batches
    the serial loop would have encoded, so the losses agree bit for bit -- eager and graph-replayed trainer alike."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import e2e
    from frlw_evd_amd.trainer import Trainer
    src = e2e.SyntheticTafSource(6, seed=77, events_per_window=8_000)
    lab = src.labels(2)
    order = [[0, 1], [2, 3], [4, 5], [0, 1]]
    serial = Trainer(e2e.build_model(16, 2), global_batch=2, nodes=1, iters_per_epoch=10)
    want = [serial.train_step(src.encode_batch(idx), lab, i)[0] for i, idx in enumerate(order)]
    for graph in (False, True):
        tr = Trainer(e2e.build_model(16, 2), global_batch=2, nodes=1, iters_per_epoch=10, graph=graph)
        # (graph: train_step captures on first use; the capture's warm-up steps do not train -- Trainer.capture restores
        #  parameters, BatchNorm buffers and Adam's state in place)
        ahead = e2e.EncodeAhead(src)
        ahead.start(order[0])
        got, started = [], []
        for i in range(len(order)):
            nxt = order[i + 1] if i + 1 < len(order) else None

            def after(nxt=nxt):
                started.append(nxt)
                if nxt is not None:
                    ahead.start(nxt)
            got.append(tr.train_step(ahead.take(), lab, i, after_launch=after)[0])
        assert started == order[1:] + [None]
        assert got == want, (graph, got, want)
    with pytest.raises(RuntimeError):
        ahead.take()


@pytest.mark.parametrize("recipe", ["yolox_taf_bfm", "yolox"])
def test_raw_files_to_checkpoint_to_evaluation(tmp_path, recipe):
    """The whole offline flow of README.md:56-170 on a fabricated GEN1 dataset, every step through the product's entry points:
    ``*_td.dat`` + ``*_bbox.npy``  ->  ``generate_taf.py`` / ``generate_eventvolume.py`` (uint8 representation files)  ->
    ``train.py --bbox_path --data_path`` (disk datasets, one epoch, checkpoints)  ->  ``test.py --record`` (summarise.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import shutil
    import socket
    import subprocess
    import sys
    import harness_data
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    raw, lab = harness_data.build(str(tmp_path / "dataset"))
    for d in (raw, lab):  # a validation split: the training sequence once more
        shutil.copytree(os.path.join(d, "train"), os.path.join(d, "val"))
    target = str(tmp_path / "processed")
    gen, sub, bins = (("generate_taf.py", "taf", "4") if recipe == "yolox_taf_bfm" else ("generate_eventvolume.py", "EventVolume250000", "5"))
    r = subprocess.run([sys.executable, os.path.join(root, gen), "-raw_dir", raw, "-label_dir", lab, "-target_dir", target,
                        "-dataset", "gen1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FRLW_MAX_EPOCHS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    log = str(tmp_path / "log") + "/"
    common = ["--dataset", "gen1", "--exp_type", recipe, "--event_volume_bins", bins, "--nodes", "1", "--log_path", log,
              "--bbox_path", lab, "--data_path", os.path.join(target, sub), "--num_cpu_workers", "2"]
    r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--batch_size", "2", "--augmentation", "True",
                        "--exp_name", "R"] + common, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "train_loader_len: 1, test_loader_len: 1" in r.stdout and "trainloss" in r.stdout
    assert os.path.exists(os.path.join(log, "R", "checkpoints", "best_epoch.pth"))
    r = subprocess.run([sys.executable, os.path.join(root, "test.py"), "--batch_size", "4", "--record", "True", "--resume_exp", "R"]
                       + common, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "test_loader_len: 2" in r.stdout and os.path.exists(os.path.join(log, "R", "summarise.npz"))  # 7 labelled frames, batch 4
