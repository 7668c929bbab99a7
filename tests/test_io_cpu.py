"""CPU: .flo round trip (byte layout of utils/flow.py:11-34) and reference-compatible checkpoints."""
import os
import types

import numpy as np
import pytest
import torch

import irr_amd
from irr_amd import io as I
from irr_amd.train import ModelAndLoss


def test_flo_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    uv = rng.standard_normal((5, 7, 2)).astype(np.float32)
    p = str(tmp_path / "a.flo")
    I.write_flow(p, uv)
    raw = open(p, "rb").read()
    assert len(raw) == 4 + 4 + 4 + 5 * 7 * 2 * 4
    assert np.frombuffer(raw[:4], np.float32)[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (7, 5)               # width first, then height
    np.testing.assert_array_equal(np.frombuffer(raw[12:], np.float32).reshape(5, 7, 2), uv)
    np.testing.assert_array_equal(I.read_flo_as_float32(p), uv)
    I.write_flow(p, uv[:, :, 0], uv[:, :, 1])
    np.testing.assert_array_equal(I.read_flo_as_float32(p), uv)
    I.flow_tensor_to_flo(p, torch.from_numpy(uv.transpose(2, 0, 1))[None])
    np.testing.assert_array_equal(I.read_flo_as_float32(p), uv)
    open(p, "wb").write(b"\\0" * 12)
    with pytest.raises(ValueError):
        I.read_flo_as_float32(p)


def test_checkpoint_reference_layout(tmp_path):
    args = types.SimpleNamespace(batch_size=2, model_div_flow=0.05)
    torch.manual_seed(1)
    m = irr_amd.PWCNet(args)
    mal = ModelAndLoss(args, m, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args))
    saver = I.CheckpointSaver()
    f = saver.save_latest(str(tmp_path), mal, {"epoch": 3, "epe": 1.5}, store_as_best=True)
    assert os.path.basename(f) == "checkpoint_latest.ckpt" and os.path.exists(str(tmp_path / "checkpoint_best.ckpt"))
    ck = torch.load(f, weights_only=False)
    assert set(ck.keys()) == {"epoch", "epe", "state_dict"}
    assert len(ck["state_dict"]) == 124 and all(k.startswith("_model.") for k in ck["state_dict"])
    # restore into a differently initialised model, excluding the refinement heads
    torch.manual_seed(2)
    m2 = irr_amd.PWCNet(args)
    mal2 = ModelAndLoss(args, m2, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args))
    before = m2.refine_flow.convs[0][0].weight.clone()
    stats, _ = saver.restore_latest(str(tmp_path), mal2, include_params="*", exclude_params=["_model.refine_*"])
    assert stats == {"epoch": 3, "epe": 1.5}
    assert torch.equal(m2.flow_estimators.conv1[0].weight, m.flow_estimators.conv1[0].weight)
    assert torch.equal(m2.refine_flow.convs[0][0].weight, before)


# ---- evaluation-path formats ----
def test_png_roundtrip_and_filters(tmp_path):
    import struct
    import zlib
    from irr_amd import io
    rng = np.random.default_rng(0)
    for dt, hi in ((np.uint8, 256), (np.uint16, 65536)):
        img = rng.integers(0, hi, size=(13, 17, 3)).astype(dt)
        fn = str(tmp_path / f"a_{np.dtype(dt).name}.png")
        io.write_png(fn, img)
        back = io.read_png(fn)
        assert back.dtype == dt and np.array_equal(back, img)
    # a PNG whose scanlines use the sub / up / average / paeth filters (as other encoders emit them) decodes identically
    img = rng.integers(0, 256, size=(8, 5, 3)).astype(np.uint8)
    rows = img.reshape(8, -1).astype(np.int32)
    bpp, stride = 3, 15
    raw = bytearray()
    for y in range(8):
        ft = (1, 2, 3, 4, 0, 4, 3, 1)[y]
        prev = rows[y - 1] if y else np.zeros(stride, np.int32)
        line = []
        for i in range(stride):
            a = rows[y, i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
            p = {0: 0, 1: a, 2: b, 3: (a + b) >> 1, 4: a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)}[ft]
            line.append((rows[y, i] - p) & 255)
        raw += bytes([ft]) + bytes(line)

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    fn = str(tmp_path / "filtered.png")
    with open(fn, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 5, 8, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    assert np.array_equal(io.read_png(fn), img)
    with pytest.raises(ValueError):
        io.write_png(str(tmp_path / "bad.png"), np.zeros((4, 4), np.uint8))


def test_kitti_flow_png_roundtrip(tmp_path):
    """utils/flow.py:37-62 / datasets/kitti_combined.py:19-34: u*64 + 2^15 as uint16, third channel = validity."""
    from irr_amd import io
    rng = np.random.default_rng(1)
    flow = (rng.standard_normal((10, 14, 2)) * 20).astype(np.float32)
    flow[0, 0] = (600.0, -600.0)                               # beyond the representable range: clipped
    mask = (rng.random((10, 14)) < 0.7).astype(np.float64)
    fn = str(tmp_path / "f.png")
    io.write_flow_png(fn, flow, mask=mask)
    raw = io.read_png(fn)
    assert raw.dtype == np.uint16 and raw[0, 0, 0] == 65535 and raw[0, 0, 1] == 0
    back, valid = io.read_png_flow(fn)
    assert valid.shape == (10, 14, 1) and np.array_equal(valid[:, :, 0], mask.astype(int))
    want = np.clip(flow.astype(np.float64) * 64 + 2 ** 15, 0, 65535).astype(np.uint16).astype(np.float64)
    want = (want - 2 ** 15) / 64.0
    want[mask == 0] = 0
    assert np.array_equal(back, want)
    assert np.abs(back[mask == 1] - flow[mask == 1])[1:].max() <= 1.0 / 64 + 1e-6     # quantisation step of the format


def test_middlebury_colour_coding_matches_reference(golden_dir):
    from irr_amd import io
    g = np.load(os.path.join(golden_dir, "flowvis.npz"))
    assert np.array_equal(io.make_color_wheel(), g["wheel"])
    rgb = io.flow_to_png_middlebury(g["flow"])
    assert rgb.dtype == np.uint8 and np.array_equal(rgb, g["rgb"])


def test_save_outputs_layout(tmp_path):
    """runtime.py:276-343: file names and the set of files per switch."""
    import types
    from irr_amd import io
    g = torch.Generator().manual_seed(2)
    out = {"flow": torch.randn(2, 2, 8, 12, generator=g), "flow_b": torch.randn(2, 2, 8, 12, generator=g),
           "occ": torch.randn(2, 1, 8, 12, generator=g), "occ_b": torch.randn(2, 1, 8, 12, generator=g)}
    ex = {"basedir": ["alley_1", "alley_1"], "basename": ["frame_0001", "frame_0002"]}
    args = types.SimpleNamespace(save=str(tmp_path / "out"), save_result_img=True, save_result_flo=True, save_result_png=True,
                                 save_result_occ=True, save_result_bidirection=True)
    files = io.save_outputs(args, ex, out)
    rel = sorted(os.path.relpath(f, args.save) for f in files)
    want = []
    for n in ("frame_0001", "frame_0002"):
        want += [f"img/alley_1/{n}_occ.png", f"img/alley_1/{n}_occ_b.png", f"img/alley_1/{n}_flow.png", f"img/alley_1/{n}_flow_b.png",
                 f"flo/alley_1/{n}.flo", f"flo/alley_1/{n}.png"]
    assert rel == sorted(want)
    assert np.array_equal(io.read_flo_as_float32(os.path.join(args.save, "flo/alley_1/frame_0002.flo")),
                          out["flow"][1].numpy().transpose(1, 2, 0))
    occ = io.read_png(os.path.join(args.save, "img/alley_1/frame_0001_occ.png"))
    assert set(np.unique(occ)) <= {0, 255} and np.array_equal(occ[:, :, 0] == 255, (torch.sigmoid(out["occ"][0, 0]) > 0.5).numpy())
    args2 = types.SimpleNamespace(save=str(tmp_path / "o2"), save_result_img=False, save_result_flo=True, save_result_png=False,
                                  save_result_occ=False, save_result_bidirection=False)
    files2 = io.save_outputs(args2, {"basename": ["a", "b"]}, out)
    assert sorted(os.path.relpath(f, args2.save) for f in files2) == ["flo/a.flo", "flo/b.flo"]
