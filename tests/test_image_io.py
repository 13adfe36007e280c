"""The files the render path leaves behind: `out.exr` of `--image` (src/main.rs:171-183, Rgba32F with alpha 1)
and PFM.  The writers are checked with a reader written here from the OpenEXR file layout (magic, version,
attribute list, line offset table, one chunk per scanline with the channels in alphabetical order)."""
import struct

import numpy as np
import pytest

import raytracing_amd as rt


def read_exr(path):
    data = open(path, "rb").read()
    assert data[:4] == bytes([0x76, 0x2F, 0x31, 0x01]), "magic"
    version, = struct.unpack_from("<I", data, 4)
    assert version == 2, "single-part scanline file, version 2, no flags"
    pos = 8
    attrs = {}

    def cstr():
        nonlocal pos
        end = data.index(b"\0", pos)
        s = data[pos:end].decode()
        pos = end + 1
        return s

    while data[pos] != 0:
        name, typ = cstr(), cstr()
        size, = struct.unpack_from("<i", data, pos)
        pos += 4
        attrs[name] = (typ, data[pos:pos + size])
        pos += size
    pos += 1
    assert attrs["compression"] == ("compression", b"\0")
    assert attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    assert attrs["displayWindow"][1] == attrs["dataWindow"][1]
    w, h = x1 - x0 + 1, y1 - y0 + 1
    # channel list
    ch, cp, names = attrs["channels"][1], 0, []
    while ch[cp] != 0:
        end = ch.index(b"\0", cp)
        names.append(ch[cp:end].decode())
        ptype, = struct.unpack_from("<i", ch, end + 1)
        xs, ys = struct.unpack_from("<2i", ch, end + 9)
        assert (ptype, xs, ys) == (2, 1, 1), "32-bit float channels without subsampling"
        cp = end + 17
    assert names == sorted(names)
    offsets = struct.unpack_from(f"<{h}Q", data, pos)
    img = np.empty((h, w, len(names)), np.float32)
    for y in range(h):
        yy, nbytes = struct.unpack_from("<2i", data, offsets[y])
        assert yy == y + y0 and nbytes == 4 * w * len(names)
        line = np.frombuffer(data, "<f4", w * len(names), offsets[y] + 8).reshape(len(names), w)
        img[y] = line.T
    assert offsets[-1] + 8 + 4 * w * len(names) == len(data)
    return {n: img[..., i] for i, n in enumerate(names)}


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0, "little endian"
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)[::-1]


def test_exr_and_pfm_hold_the_pixels(tmp_path):
    rng = np.random.default_rng(5)
    img = rng.standard_normal((7, 13, 4)).astype(np.float32)
    img[0, 0] = (0.0, np.inf, -0.0, 1.0)
    rt.write_exr(str(tmp_path / "a.exr"), img)
    ch = read_exr(str(tmp_path / "a.exr"))
    assert list(ch) == ["A", "B", "G", "R"]
    for i, n in enumerate("RGBA"):
        assert np.array_equal(ch[n].view(np.uint32), img[..., i].view(np.uint32)), n
    rt.write_pfm(str(tmp_path / "a.pfm"), img)
    assert np.array_equal(read_pfm(str(tmp_path / "a.pfm")).view(np.uint32), img[..., :3].view(np.uint32))
    with pytest.raises(rt.RenderError):
        rt.write_exr(str(tmp_path / "no_such_dir" / "a.exr"), img)


@pytest.mark.gpu
def test_render_to_image_is_the_image_mode_of_the_reference(tmp_path):
    """sample_count - 1 empty renders + one hdr render; the file holds sqrt(mean) with alpha 1."""
    a = rt.Scene.builtin(6, 48, 32)
    img = a.render_to_image(5, str(tmp_path / "out.exr"))
    b = rt.Scene.builtin(6, 48, 32)
    for _ in range(4):
        b.empty_render()
    hdr = b.hdr_render()
    assert a.samples == b.samples == 5
    assert np.array_equal(img[..., :3], hdr) and (img[..., 3] == 1.0).all()
    assert np.array_equal(hdr, np.sqrt(b.accumulator()[..., :3] * np.float32(1.0 / 5)))
    ch = read_exr(str(tmp_path / "out.exr"))
    assert np.array_equal(np.stack([ch["R"], ch["G"], ch["B"], ch["A"]], -1), img)
    a.render_to_image(1, str(tmp_path / "out.pfm"))
    assert read_pfm(str(tmp_path / "out.pfm")).shape == (32, 48, 3)


@pytest.mark.gpu
def test_checkpoint_resumes_the_same_render(tmp_path):
    whole = rt.Scene.builtin(8, 40, 24)
    whole.set_seed(11)
    whole.render_samples(7)
    first = rt.Scene.builtin(8, 40, 24)
    first.set_seed(11)
    first.render_samples(3)
    first.save_checkpoint(str(tmp_path / "ck.bin"))
    second = rt.Scene.builtin(8, 40, 24)
    second.load_checkpoint(str(tmp_path / "ck.bin"))
    assert second.samples == 3
    second.render_samples(4)
    assert np.array_equal(whole.accumulator(), second.accumulator())
    other = rt.Scene.builtin(8, 41, 24)
    with pytest.raises(rt.RenderError):
        other.load_checkpoint(str(tmp_path / "ck.bin"))
    # sums of another image must not be continued: a camera that has turned, another scene of the same size
    turned = rt.Scene.builtin(8, 40, 24)
    turned.change_pitch_yaw_by(0.0, 5.0)
    with pytest.raises(rt.RenderError, match="another camera pose"):
        turned.load_checkpoint(str(tmp_path / "ck.bin"))
    with pytest.raises(rt.RenderError, match="another camera pose or another scene"):
        rt.Scene.builtin(6, 40, 24).load_checkpoint(str(tmp_path / "ck.bin"))
    # the file is written beside the target and moved into place: nothing else is left behind, and a second save
    # replaces the first
    first.render_samples(1)
    first.save_checkpoint(str(tmp_path / "ck.bin"))
    assert sorted(p.name for p in tmp_path.iterdir()) == ["ck.bin"]
    third = rt.Scene.builtin(8, 40, 24)
    third.load_checkpoint(str(tmp_path / "ck.bin"))
    assert third.samples == 4
