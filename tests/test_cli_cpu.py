"""CPU tests (-m "not gpu") of the batch driver's host logic (cvsteer_amd/run.py): file lists, image
decoding, rank sharding -- everything that does not need the engine."""
import os

import numpy as np

from cvsteer_amd import run
from cvsteer_amd.batch import shard_range


def test_input_list_semantics(tmp_path):
    # example/steer.cpp:156-165: ".txt" or no extension -> list of files; anything else -> one image
    lst = tmp_path / "files.txt"
    lst.write_text("a.png\n\n  b.jpg  \nc.npy\n")
    assert run.input_list(str(lst)) == ["a.png", "b.jpg", "c.npy"]
    noext = tmp_path / "listing"
    noext.write_text("x.png\n")
    assert run.input_list(str(noext)) == ["x.png"]
    assert run.input_list("/some/where/img.png") == ["/some/where/img.png"]


def test_read_gray_and_write_u8(tmp_path, golden_dir):
    fish = run.read_gray(os.path.join(golden_dir, "fish.jpg"))
    assert fish.dtype == np.uint8 and fish.shape == (185, 256)
    assert np.array_equal(fish, np.load(os.path.join(golden_dir, "fish_u8.npy")))
    rgb = np.stack([fish, fish, fish], axis=-1)
    np.save(tmp_path / "rgb.npy", rgb)
    g = run.read_gray(str(tmp_path / "rgb.npy"))
    assert g.shape == fish.shape and np.abs(g - fish.astype(np.float32)).max() < 1e-3   # 0.299+0.587+0.114 = 1
    for ext in (".png", ".npy"):
        p = str(tmp_path / ("o" + ext))
        run.write_u8(p, fish)
        assert np.array_equal(run.read_gray(p), fish)


def test_files_shard_over_ranks_like_frames():
    files = ["f%03d" % i for i in range(37)]
    seen = []
    for r in range(8):
        lo, hi = shard_range(len(files), 8, r)
        seen += files[lo:hi]
    assert seen == files
