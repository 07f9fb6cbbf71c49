"""The disk-backed datasets (frlw_evd_amd/dataset.py) against the reference's own ``propheseeDataset`` /
``propheseeTafDataset`` / ``collate_events`` run on the same fabricated directory (tests/golden/dataset_files.npz,
make_golden_dataset_files.py): sample lists, volumes, labels on the CPU; the loader's GPU batches in the ``gpu`` tests."""
import os

import numpy as np
import pytest

import dataset_fixture as fx
from frlw_evd_amd import dataset as ds


@pytest.fixture(scope="module")
def dirs(tmp_path_factory):
    return fx.build(str(tmp_path_factory.mktemp("dsfiles")))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "dataset_files.npz"))


def _sorted(d):
    return np.argsort([f"{n}_{int(t):012d}" for n, t in zip(d.file_name, d.sequence_end_t)])


def _make(kind, dirs, mode, **kw):
    bbox, ev, taf = dirs
    if kind == "ev":
        return ds.propheseeDataset(bbox, ev, "gen1", fx.IMG, fx.IMG, fx.BINS, 10000, 1, mode, False, False, **kw)
    return ds.propheseeTafDataset(bbox, taf, "gen1", fx.IMG, fx.IMG, 10000, int(kind[3:]), mode, False, False)


@pytest.mark.parametrize("mode", ["train", "val", "test"])
@pytest.mark.parametrize("kind", ["ev", "taf8", "taf4"])
def test_sample_lists_volumes_and_labels(dirs, gold, kind, mode):
    d = _make(kind, dirs, mode, **({"reference_mean_quirk": True} if kind == "ev" else {}))
    order = _sorted(d)
    assert [d.file_name[i] for i in order] == list(gold[f"{kind}_{mode}_names"])          # timestamps without a file are skipped
    assert [int(d.sequence_end_t[i]) for i in order] == list(gold[f"{kind}_{mode}_times"])
    vols = np.stack([d.load_data(int(i)) for i in order])
    assert vols.dtype == np.float32 and vols.tobytes() == gold[f"{kind}_{mode}_load_data"].tobytes()
    labels = np.stack([d.labels(int(i))[0] for i in order])
    assert labels.tobytes() == gold[f"{kind}_{mode}_labels"].tobytes()
    if kind == "ev":
        # the product's default loads ALL 2 * bins channels (the HEAD line leaves two: data/dataset.py:245, see dataset.py)
        full = _make("ev", dirs, mode)
        assert full.channels == 2 * fx.BINS and full.load_data(int(_sorted(full)[0])).shape == (2 * fx.BINS, *fx.IMG)
        u8 = full.load_u8(int(_sorted(full)[0]))
        assert u8.dtype == np.uint8 and np.array_equal(u8.astype(np.float32).mean(0), vols[0][0])


def test_getitem_keeps_uint8(dirs):
    d = _make("taf8", dirs, "train")
    vol, labels, params, name, t = d[0]
    assert vol.dtype == np.uint8 and vol.shape == (2 * fx.K, *fx.IMG) and labels.shape == (80, 5)
    assert params.sr == 1.0 and not params.flip and name in fx.SEQS["train"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ev", "taf8", "taf4"])
def test_loader_batches_on_the_gpu(dirs, gold, kind):
    """``Loader``: uint8 over PCIe, ``/255`` + zoom + crop + flip as one kernel on the batch -- the images the reference's
    ``__getitem__`` + ``collate_events`` produce on the CPU, bit for bit (no augmentation in the validation split)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _make(kind, dirs, "val", **({"reference_mean_quirk": True} if kind == "ev" else {}))
    order = [int(i) for i in _sorted(d)]
    loader = ds.Loader(d, batch_size=8, num_workers=2, pin_memory=True, device="cuda", shuffle=False, sampler=order)
    assert len(loader) == 1
    (imgs, labels, names, stamps), = list(loader)
    assert imgs.is_cuda and imgs.dtype == torch.float32 and labels.is_cuda and labels.dtype == torch.float64
    assert imgs.cpu().numpy().tobytes() == gold[f"{kind}_val_img"].tobytes()
    assert labels.cpu().numpy().tobytes() == gold[f"{kind}_val_labels"].tobytes()
    if kind == "ev":
        assert list(names) == list(gold["ev_val_batch_names"]) and list(stamps) == list(gold["ev_val_batch_times"])
        assert imgs.cpu().numpy().tobytes() == gold["ev_val_batch_img"].tobytes()


@pytest.mark.gpu
def test_loader_epoch_shapes_and_prefetch(dirs):
    """Short last batch (drop_last=False), shuffling, several reader threads, augmentation on: every sample exactly once."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    bbox, ev, taf = dirs
    d = ds.propheseeTafDataset(bbox, taf, "gen1", fx.IMG, fx.IMG, 10000, 8, "train", True, False)
    loader = ds.Loader(d, batch_size=3, num_workers=4, pin_memory=True, device="cuda", shuffle=True)
    seen = []
    for imgs, labels, names, stamps in loader:
        assert imgs.shape[1:] == (16, *fx.IMG, 1, 1) and imgs.shape[0] == labels.shape[0] == len(names) == len(stamps) <= 3
        assert float(imgs.min()) >= 0.0 and float(imgs.max()) <= 1.0
        seen += [(n, int(t)) for n, t in zip(names, stamps)]
    assert len(loader) == 2 and sorted(seen) == sorted((n, int(t)) for n, t in zip(d.file_name, d.sequence_end_t))
