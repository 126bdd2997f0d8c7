"""I/O + preprocessing adapters around the pipeline: what scripts/inference_video.py needs besides the
models, without the wheels this image lacks (omegaconf, PyAV, cv2, torchvision, scikit-image).

  load_config            OmegaConf.load(path) as the script uses it (scripts/inference_video.py:64,88-89): attribute access,
                         nested mappings, `to_container`
  read_frames / get_fps  src/utils/util.py:114-137 (PyAV there).  Here: a directory of frame images, .gif / .webp / .apng
                         animations, .npy / .npz frame arrays, and .mp4 / .mov files whose video track is Motion-JPEG or PNG
                         (which is what `save_videos_grid` below writes) are decoded natively; other codecs (H.264 ...) go to
                         PyAV / cv2 / imageio when one of them is importable and raise a clear error otherwise
  frames_to_tensor       `get_tensor` of the script (:48-58): transforms.Resize((h, w)) + ToTensor, stacked to (1, c, f, h, w)
  make_grid              torchvision.utils.make_grid (nrow, padding 2, pad_value 0) restated
  save_videos_grid       src/utils/util.py:90-111: b c t h w -> one grid image per frame -> .mp4 / .gif.  The .mp4 is a plain
                         ISO-BMFF file with a Motion-JPEG ('jpeg') video track written by the ~80-line muxer below (the reference
                         uses cv2's mp4v encoder; there is no video encoder in this image, PIL's JPEG codec is); .gif via PIL
  save_videos_from_pil   src/utils/util.py:50-87
  resize_depth           skimage.transform.resize(depth, (1, H/8, W/8)) as called at scripts/inference_video.py:184: order-1
                         interpolation, 'reflect' boundary, Gaussian anti-aliasing when shrinking, clipped to the input range
                         (restated from the published algorithm on scipy.ndimage: third party, PARITY UNPINNED)
"""
import io
import math
import os
import struct
from fractions import Fraction
from pathlib import Path

import numpy as np
import torch
from PIL import Image


# ------------------------------------------------------------------------------------------------ config
class AttrConfig(dict):
    """Mapping with attribute access, nested like an OmegaConf DictConfig (read-only use: the script only reads)."""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"Missing key {name}") from None

    def __setattr__(self, name, value):
        self[name] = _wrap(value)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, AttrConfig):
        return AttrConfig(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


def load_config(path):
    import yaml
    with open(path) as fh:
        return AttrConfig(yaml.safe_load(fh) or {})


def to_container(cfg):
    """OmegaConf.to_container: plain dicts / lists."""
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [to_container(v) for v in cfg]
    return cfg


# ------------------------------------------------------------------------------------------------ ISO-BMFF (mp4) Motion-JPEG
def _box(kind, payload):
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind, version, flags, payload):
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def write_mjpeg_mp4(pil_images, path, fps=8, quality=95):
    """Minimal ISO base-media file: ftyp, mdat (one JPEG per frame), moov with ONE video track ('jpeg' sample entry)."""
    fps = Fraction(fps).limit_denominator(1000)
    timescale, delta = fps.numerator * 1000 // math.gcd(fps.numerator * 1000, fps.denominator), None
    delta = timescale * fps.denominator // fps.numerator
    width, height = pil_images[0].size
    frames = []
    for im in pil_images:
        buf = io.BytesIO()
        im.convert("RGB").save(buf, format="JPEG", quality=quality, subsampling=0)
        frames.append(buf.getvalue())
    n = len(frames)
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomiso2mp41")
    mdat = _box(b"mdat", b"".join(frames))
    first = len(ftyp) + 8
    offsets, o = [], first
    for f in frames:
        offsets.append(o)
        o += len(f)
    dur = delta * n
    ident = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, dur) + struct.pack(">IH", 0x10000, 0x100) + b"\0" * 10 + ident
                 + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) + ident
                 + struct.pack(">II", width << 16, height << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, dur, 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    entry = (b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HHIIIH", width, height, 0x480000, 0x480000, 0, 1)
             + bytes([10]) + b"mikudance\0".ljust(31, b"\0") + struct.pack(">Hh", 0x18, -1))
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"jpeg", entry))
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(f)) for f in frames))
    stco = _full(b"stco", 0, 0, struct.pack(">I", n) + b"".join(struct.pack(">I", x) for x in offsets))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\0" * 8) + dinf + stbl)
    trak = _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf))
    moov = _box(b"moov", mvhd + trak)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(ftyp + mdat + moov)


def _children(buf, start, end):
    out, p = [], start
    while p + 8 <= end:
        size, kind = struct.unpack(">I4s", buf[p:p + 8])
        hdr = 8
        if size == 1:
            size, hdr = struct.unpack(">Q", buf[p + 8:p + 16])[0], 16
        elif size == 0:
            size = end - p
        if size < hdr:
            break
        out.append((kind, p + hdr, p + size))
        p += size
    return out


def _find(buf, start, end, *path):
    for kind, a, b in _children(buf, start, end):
        if kind == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, *path[1:])
    return None


def _video_track(buf):
    moov = _find(buf, 0, len(buf), b"moov")
    if moov is None:
        raise ValueError("no moov box: not an ISO base-media (mp4 / mov) file")
    for kind, a, b in _children(buf, *moov):
        if kind != b"trak":
            continue
        hd = _find(buf, a, b, b"mdia", b"hdlr")
        if hd and buf[hd[0] + 8:hd[0] + 12] == b"vide":
            return a, b
    raise ValueError("no video track")


def _mp4_index(buf):
    """-> (codec fourcc, [(offset, size)], fps Fraction)."""
    a, b = _video_track(buf)
    stbl = _find(buf, a, b, b"mdia", b"minf", b"stbl")
    mdhd = _find(buf, a, b, b"mdia", b"mdhd")
    ver = buf[mdhd[0]]
    timescale = struct.unpack(">I", buf[mdhd[0] + (20 if ver == 1 else 12):mdhd[0] + (24 if ver == 1 else 16)])[0]
    boxes = {k: (x, y) for k, x, y in _children(buf, *stbl)}
    sd = boxes[b"stsd"][0]
    codec = buf[sd + 12:sd + 16]
    sz = boxes[b"stsz"][0]
    uniform, n = struct.unpack(">II", buf[sz + 4:sz + 12])
    sizes = [uniform] * n if uniform else list(struct.unpack(f">{n}I", buf[sz + 12:sz + 12 + 4 * n]))
    if b"stco" in boxes:
        co = boxes[b"stco"][0]
        nc = struct.unpack(">I", buf[co + 4:co + 8])[0]
        chunks = list(struct.unpack(f">{nc}I", buf[co + 8:co + 8 + 4 * nc]))
    else:
        co = boxes[b"co64"][0]
        nc = struct.unpack(">I", buf[co + 4:co + 8])[0]
        chunks = list(struct.unpack(f">{nc}Q", buf[co + 8:co + 8 + 8 * nc]))
    sc = boxes[b"stsc"][0]
    ne = struct.unpack(">I", buf[sc + 4:sc + 8])[0]
    runs = [struct.unpack(">III", buf[sc + 8 + 12 * i:sc + 20 + 12 * i]) for i in range(ne)]
    samples, si = [], 0
    for ci, off in enumerate(chunks, start=1):
        per = [r for r in runs if r[0] <= ci][-1][1]
        for _ in range(per):
            if si >= n:
                break
            samples.append((off, sizes[si]))
            off += sizes[si]
            si += 1
    tt = boxes[b"stts"][0]
    nt = struct.unpack(">I", buf[tt + 4:tt + 8])[0]
    ent = [struct.unpack(">II", buf[tt + 8 + 8 * i:tt + 16 + 8 * i]) for i in range(nt)]
    total = sum(c * d for c, d in ent)
    fps = Fraction(timescale * sum(c for c, _ in ent), total) if total else Fraction(0)
    return codec, samples, fps


_INTRA = (b"jpeg", b"mjpa", b"mjpb", b"MJPG", b"mjpg", b"png ")


def _external_video(path, want):
    """H.264 & co: whatever decoder package is importable."""
    try:
        import av
        container = av.open(path)
        stream = next(s for s in container.streams if s.type == "video")
        if want == "fps":
            return Fraction(stream.average_rate)
        return [frame.to_image().convert("RGB") for packet in container.demux(stream) for frame in packet.decode()]
    except ImportError:
        pass
    try:
        import cv2
        cap = cv2.VideoCapture(path)
        if want == "fps":
            return Fraction(cap.get(cv2.CAP_PROP_FPS)).limit_denominator(1000)
        out = []
        while True:
            ok, fr = cap.read()
            if not ok:
                return out
            out.append(Image.fromarray(cv2.cvtColor(fr, cv2.COLOR_BGR2RGB)))
    except ImportError:
        pass
    try:
        import imageio.v3 as iio
        if want == "fps":
            return Fraction(iio.immeta(path).get("fps", 0)).limit_denominator(1000)
        return [Image.fromarray(f).convert("RGB") for f in iio.imiter(path)]
    except ImportError:
        pass
    raise RuntimeError(f"{path}: its video codec needs an external decoder and none of PyAV / cv2 / imageio is installed.  Pass a "
                       "directory of frame images, a .gif / .webp / .npy, or a Motion-JPEG .mp4 (what save_videos_grid writes).  On any "
                       "machine that has one of those packages, `python -m mikudance_amd.io_utils convert <clip>.mp4 <clip>.mjpeg.mp4` "
                       "(or `... <clip>_frames/`) re-encodes the clip once into a form this reader decodes natively")


_IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")


def read_frames(video_path):
    """src/utils/util.py:114-128 -> list of RGB PIL images."""
    p = str(video_path)
    if os.path.isdir(p):
        names = sorted(f for f in os.listdir(p) if f.lower().endswith(_IMG_EXT))
        if not names:
            raise ValueError(f"{p}: no frame images")
        return [Image.open(os.path.join(p, f)).convert("RGB") for f in names]
    ext = Path(p).suffix.lower()
    if ext in (".npy", ".npz"):
        arr = np.load(p)
        arr = arr[arr.files[0]] if ext == ".npz" else arr
        return [Image.fromarray(np.asarray(f, dtype=np.uint8)).convert("RGB") for f in arr]
    if ext in (".gif", ".webp", ".apng", ".png"):
        im = Image.open(p)
        out = []
        for i in range(getattr(im, "n_frames", 1)):
            im.seek(i)
            out.append(im.convert("RGB"))
        return out
    buf = open(p, "rb").read()
    codec, samples, _ = _mp4_index(buf)
    if codec in _INTRA:
        return [Image.open(io.BytesIO(buf[o:o + s])).convert("RGB") for o, s in samples]
    return _external_video(p, "frames")


def get_fps(video_path):
    """src/utils/util.py:131-137 (stream.average_rate, a Fraction)."""
    p = str(video_path)
    ext = Path(p).suffix.lower()
    if os.path.isdir(p) or ext in (".npy", ".npz"):
        return Fraction(8)
    if ext in (".gif", ".webp", ".apng", ".png"):
        d = Image.open(p).info.get("duration", 125) or 125
        return Fraction(1000, int(d))
    buf = open(p, "rb").read()
    codec, _, fps = _mp4_index(buf)
    return fps if fps else _external_video(p, "fps")


# ------------------------------------------------------------------------------------------------ tensors <-> images
def frames_to_tensor(pils, height, width):
    """`get_tensor` (scripts/inference_video.py:48-58): Resize((h, w)) [bilinear] + ToTensor per frame -> (1, c, f, h, w)."""
    frames = [torch.from_numpy(np.asarray(im.convert("RGB").resize((width, height), Image.BILINEAR), dtype=np.float32) / 255.0)
              for im in pils]
    return torch.stack(frames, 0).permute(3, 0, 1, 2)[None].contiguous()


def make_grid(tensor, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a (B, C, H, W) batch: one image is returned as is; otherwise images are laid out `nrow`
    per row on a canvas of pad_value with `padding` pixels around and between them."""
    if tensor.dim() != 4:
        raise ValueError("make_grid expects (B, C, H, W)")
    if tensor.shape[1] == 1:
        tensor = tensor.expand(-1, 3, -1, -1)
    nmaps = tensor.shape[0]
    if nmaps == 1:
        return tensor[0]
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(nmaps / xmaps))
    h, w = tensor.shape[2] + padding, tensor.shape[3] + padding
    grid = tensor.new_full((tensor.shape[1], h * ymaps + padding, w * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid[:, y * h + padding:(y + 1) * h, x * w + padding:(x + 1) * w] = tensor[k]
            k += 1
    return grid


def save_videos_from_pil(pil_images, path, fps=8):
    """src/utils/util.py:50-87 / :140-175: .mp4 or .gif."""
    fmt = Path(path).suffix.lower()
    if fmt == ".mp4":
        write_mjpeg_mp4(pil_images, path, fps=fps)
    elif fmt == ".gif":
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        pil_images[0].save(fp=path, format="GIF", append_images=pil_images[1:], save_all=True, duration=(1 / float(fps) * 1000), loop=0)
    else:
        raise ValueError("Unsupported file type. Use .mp4 or .gif.")


def save_videos_grid(videos, path, rescale=False, n_rows=6, fps=8):
    """src/utils/util.py:90-111: videos (b, c, t, h, w) in [0, 1] ([-1, 1] with rescale) -> one grid frame per t."""
    videos = videos.permute(2, 0, 1, 3, 4)
    outputs = []
    for x in videos:
        x = make_grid(x, nrow=n_rows).permute(1, 2, 0)
        if rescale:
            x = (x + 1.0) / 2.0
        outputs.append(Image.fromarray((x * 255).numpy().astype(np.uint8)))
    save_videos_from_pil(outputs, path, fps)


def resize_depth(depth_map, output_shape):
    """skimage.transform.resize(image, output_shape) with its defaults for a float image (order=1, mode='reflect',
    anti_aliasing=True, clip=True, preserve_range=False), the call of scripts/inference_video.py:184 on the (1, H, W) depth map:
    Gaussian pre-filter with sigma = max(0, (factor - 1) / 2) per axis (boundary 'mirror' = skimage's 'reflect'), then
    scipy.ndimage.zoom(order 1, grid_mode) and a clip to the input's range."""
    from scipy import ndimage as ndi
    image = np.asarray(depth_map, dtype=np.float64)
    output_shape = tuple(int(s) for s in output_shape)
    if image.ndim != len(output_shape):
        raise ValueError("output_shape must have one entry per input dimension")
    if image.shape == output_shape:
        return image.copy()
    factors = np.divide(image.shape, output_shape)
    filtered = image
    if np.any(factors > 1):
        sigma = np.maximum(0, (factors - 1) / 2)
        filtered = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")
    out = ndi.zoom(filtered, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)
    return np.clip(out, image.min(), image.max())


# ------------------------------------------------------------------------------------------------ one-time conversion of codec'd clips
def convert_video(src, dst, quality=95):
    """Re-encode `src` (anything read_frames can decode on THIS machine: H.264 needs PyAV / cv2 / imageio here) into a form read_frames
    decodes natively everywhere: `dst` ending in .mp4 -> Motion-JPEG mp4 at the source's frame rate; anything else -> a directory of
    numbered PNG frames.  The reference's demo clips (demo_samples/poses/*.mp4, H.264) need this once before the drop-in script can read
    them on a box without a video decoder: point tgt_pose_path / tgt_face_path / tgt_hand_path of configs/inference/inference_video.yaml at
    the converted files (reference src/utils/util.py:106-137 reads through PyAV).  Returns the number of frames written."""
    frames = read_frames(src)
    if str(dst).lower().endswith(".mp4"):
        write_mjpeg_mp4(frames, str(dst), fps=get_fps(src) or 8, quality=quality)
    else:
        os.makedirs(str(dst), exist_ok=True)
        for i, im in enumerate(frames):
            im.save(os.path.join(str(dst), f"{i:06d}.png"))
    return len(frames)


if __name__ == "__main__":
    import sys
    if len(sys.argv) != 4 or sys.argv[1] != "convert":
        sys.exit("usage: python -m mikudance_amd.io_utils convert <source video> <target .mp4 (Motion-JPEG) | target directory (PNG frames)>")
    print(f"{convert_video(sys.argv[2], sys.argv[3])} frames -> {sys.argv[3]}")
