"""Training data path (genima_amd/data.py) against the reference's reader and pre-processing semantics
(diffusion/rlbench_dataset/rlbench_dataset.py:70-210; diffusion/train_controlnet_genima.py:870-964) on a synthetic RLBench tree."""
import os
import pickle

import numpy as np
import torch
from PIL import Image

from genima_amd import data as D
from genima_amd.pipeline import HashTokenizer


def _png(path, arr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def _tree(root, n_eps=3, n_frames=5, size=(40, 56)):
    rng = np.random.RandomState(0)
    for task in ("open_box", "close_jar"):
        base = os.path.join(root, task, "variation0")
        os.makedirs(os.path.join(base, "episodes"), exist_ok=True)
        with open(os.path.join(base, "variation_descriptions.pkl"), "wb") as f:
            pickle.dump([f"{task} a", f"{task} b"], f)
        for e in (0, 1, 10, 2)[: n_eps + 1]:  # episode10 must sort after episode2 (natsort)
            for kind in ("rgb", "rgb_rendered", "front_rgb", "front_rgb_rendered"):
                for i in range(n_frames):
                    _png(os.path.join(base, "episodes", f"episode{e}", kind, f"{i}.png"), rng.randint(0, 256, size + (3,), dtype=np.uint8))
    return root


def test_reader_order_quirks_and_len(tmp_path):
    root = _tree(str(tmp_path))
    ds = D.RLBenchDataset(root, tasks="open_box,close_jar", num_demos=3)
    # 2 tasks x 3 demos (natural order 0, 1, 2 -- episode10 is cut by num_demos) x (5 - 1) frames: the tiled reader drops the last frame
    assert len(ds) == 2 * 3 * 4
    eps = [os.path.basename(os.path.dirname(os.path.dirname(e["image"]))) for e in ds.examples[:12:4]]
    assert eps == ["episode0", "episode1", "episode2"]
    assert all(e["text"] == "tiled perspectives of a robot " for e in ds.examples)  # the reference's truncated caption (Appendix F.1)
    assert ds.examples[3]["image"].endswith("rgb_rendered/3.png") and ds.examples[3]["conditioning_image"].endswith("rgb/3.png")
    fut = D.RLBenchDataset(root, tasks="open_box", num_demos=1, predict_future=True, predict_future_horizon=2)
    assert [os.path.basename(e["image"]) for e in fut.examples] == ["2.png", "3.png", "3.png", "3.png"]
    cam = D.RLBenchDataset(root, tasks="open_box", num_demos=1, tiled=False, cameras="front")
    assert len(cam) == 5 and cam.examples[0]["text"] == "a robot arm executing '" and "front_rgb_rendered" in cam.examples[0]["image"]
    ex = ds[0]
    assert ex["image"]["bytes"][:4] == b"\x89PNG" and ex["image"]["path"] == ds.examples[0]["image"]


def test_preprocess_and_collate_match_the_reference_formulas(tmp_path):
    root = _tree(str(tmp_path), n_eps=1)
    ds = D.RLBenchDataset(root, tasks="open_box", num_demos=1)
    R = 32
    ex = [ds[0], ds[1]]
    u8 = D.resize_center_crop_u8(ex[0]["image"], R)
    # torchvision Resize(R) on a 56 x 40 (W x H) image: shorter side H -> R, W -> int(R * 56 / 40) = 44; CenterCrop(R): left = round(6) = 6
    ref = Image.open(ds.examples[0]["image"]).convert("RGB").resize((44, 32), Image.BILINEAR).crop((6, 0, 38, 32))
    assert u8.shape == (R, R, 3) and np.array_equal(u8, np.asarray(ref))
    tok = HashTokenizer(1024)
    b = D.collate_fn(ex, tok, R)
    assert b["pixel_values"].shape == (2, 3, R, R) and b["pixel_values"].dtype == torch.float32 and b["input_ids"].shape == (2, 77)
    want = (torch.from_numpy(u8).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5  # ToTensor + Normalize([0.5], [0.5])
    assert torch.equal(b["pixel_values"][0], want)
    cu8 = D.resize_center_crop_u8(ex[0]["conditioning_image"], R)
    assert torch.equal(b["conditioning_pixel_values"][0], torch.from_numpy(cu8).permute(2, 0, 1).float() / 255.0)
    assert float(b["conditioning_pixel_values"].min()) >= 0.0 and float(b["pixel_values"].min()) >= -1.0
    ub = D.collate_u8(ex, tok, R)
    assert ub["pixel_values_u8"].dtype == torch.uint8 and tuple(ub["pixel_values_u8"].shape) == (2, R, R, 3)


def test_loader_epochs_shards_and_prefetch(tmp_path):
    root = _tree(str(tmp_path))
    ds = D.RLBenchDataset(root, tasks="open_box,close_jar", num_demos=3)
    tok = HashTokenizer(1024)
    full = D.DataLoader(ds, 5, tok, 16, shuffle=True, seed=3, prefetch=2)
    assert len(full) == 5  # 24 examples in batches of 5: the last partial batch is kept (DataLoader default drop_last=False)
    sizes = [b["pixel_values_u8"].shape[0] for b in full]
    assert sizes == [5, 5, 5, 5, 4]
    a = D.DataLoader(ds, 4, tok, 16, seed=3, prefetch=0)._batches()
    b0, b1 = (D.DataLoader(ds, 4, tok, 16, seed=3, rank=r, world=2)._batches() for r in range(2))
    flat = lambda bs: [i for b in bs for i in b]  # noqa: E731
    assert sorted(flat(b0) + flat(b1)) == list(range(24)) and not set(flat(b0)) & set(flat(b1))  # ranks see disjoint halves
    assert sorted(flat(a)) == list(range(24)) and flat(a) != list(range(24))
    # ranks always run the same number of FULL batches (accelerate even_batches: the tail wraps round to the start of the permutation);
    # a rank that ran out early would leave the others waiting in the gradient exchange
    for n_items, world, bs in ((10, 4, 1), (24, 3, 5), (24, 2, 4), (7, 8, 2)):
        loaders = [D.DataLoader(list(range(n_items)), bs, tok, 16, seed=1, rank=r, world=world) for r in range(world)]
        per_rank = [ld._batches() for ld in loaders]
        assert len({len(b) for b in per_rank}) == 1 and all(len(b) == len(ld) for b, ld in zip(per_rank, loaders))
        assert all(len(x) == bs for b in per_rank for x in b)
        assert set(flat([x for b in per_rank for x in b])) == set(range(n_items))
    x = [t["input_ids"].clone() for t in D.DataLoader(ds, 8, tok, 16, shuffle=False, prefetch=0)]
    y = [t["input_ids"].clone() for t in D.DataLoader(ds, 8, tok, 16, shuffle=False, prefetch=3)]
    assert all(torch.equal(p, q) for p, q in zip(x, y)) and len(x) == len(y) == 3
