"""On-disk formats of the reference, without `plyfile` (absent in this image):

* `point_cloud.ply` written by scene/gaussian_model.py:356-406 (`save_ply`) and read back by :408-470 (`load_ply`): binary
  little-endian, ONE `vertex` element, 21 `float` properties in the order of `GAUSSIAN_PLY_PROPERTIES`, all raw / pre-activation.
* the initial point cloud `point_cloud_{dense|sfm}.ply` of utils/ply_utils.py:5-34: `x y z red green blue` (uchar colours are
  scaled by 1/255 on read; `save_ply` there writes ASCII floats).
* `transforms_{train,test}.json` (dataset/blender_dataset.py:30-75): `camera_angle_x` + frames sorted by `file_path`, each with a
  camera-to-world `transform_matrix` that is converted from Blender (Y up, Z back) to COLMAP axes and inverted.
* `cfg.json` = `vars(Config)` (train.py:68-69).

Host-side Python like the reference's own loaders; numpy only. The PLY reader handles ascii / binary little / big endian files
whose first element has scalar properties (what both writers above and plyfile produce); later elements are ignored.
"""
import json
import math
import os

import numpy as np

GAUSSIAN_PLY_PROPERTIES = ("x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3",
                           "normal_0", "normal_1", "normal_2", "roughness", "f0_0", "f0_1", "f0_2")
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4",
              "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
_PLY_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def read_ply(path):
    """Returns (structured numpy array of the first element, element name)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    if len(elements) == 1:
                        raise ValueError(f"{path}: list properties in the first element are not supported")
                    elements[-1][2].append(None)
                else:
                    if tok[1] not in _PLY_TYPES:
                        raise ValueError(f"{path}: unknown property type {tok[1]}")
                    elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian") or not elements:
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
        name, count, props = elements[0]
        if fmt == "ascii":
            dtype = np.dtype([(n, t) for n, t in props])
            rows = np.loadtxt(f, max_rows=count, ndmin=2, dtype=np.float64) if count else np.zeros((0, len(props)))
            if rows.shape != (count, len(props)):
                raise ValueError(f"{path}: expected {count} x {len(props)} values, found {rows.shape}")
            out = np.empty(count, dtype)
            for k, (n, _) in enumerate(props):
                out[n] = rows[:, k]
            return out, name
        order = "<" if fmt == "binary_little_endian" else ">"
        dtype = np.dtype([(n, order + t) for n, t in props])
        data = np.fromfile(f, dtype=dtype, count=count)
        if len(data) != count:
            raise ValueError(f"{path}: truncated PLY body ({len(data)} of {count} vertices)")
        return data.astype(dtype.newbyteorder("=")), name


def write_ply(path, columns, text=False, element="vertex"):
    """columns: ordered {property name: 1-D array}; dtypes are kept (float32 -> `float`, uint8 -> `uchar`, ...)."""
    names = list(columns)
    arrays = [np.ascontiguousarray(columns[n]) for n in names]
    n = len(arrays[0]) if arrays else 0
    codes = [a.dtype.kind + str(a.dtype.itemsize) for a in arrays]
    header = ["ply", "format ascii 1.0" if text else "format binary_little_endian 1.0", f"element {element} {n}"]
    header += [f"property {_PLY_NAMES[c]} {name}" for c, name in zip(codes, names)] + ["end_header"]
    rec = np.empty(n, np.dtype([(name, "<" + c) for name, c in zip(names, codes)]))
    for name, a in zip(names, arrays):
        rec[name] = a
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        if text:
            for row in rec:
                f.write((" ".join(repr(x.item()) if x.dtype.kind == "f" else str(x.item()) for x in row) + "\n").encode("ascii"))
        else:
            rec.tofile(f)


def save_gaussians_ply(path, g):
    """scene/gaussian_model.py:356-406. g: dict of RAW parameters mean[N,3] rgb[N,3] opacity[N,1] scale[N,3] rotation[N,4]
    normal[N,3] roughness[N,1] f0[N,3] (numpy or torch)."""
    a = lambda x: np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, np.float32)
    flat = np.concatenate([a(g["mean"]), a(g["rgb"]), a(g["opacity"]).reshape(-1, 1), a(g["scale"]), a(g["rotation"]), a(g["normal"]),
                           a(g["roughness"]).reshape(-1, 1), a(g["f0"])], axis=1)
    assert flat.shape[1] == len(GAUSSIAN_PLY_PROPERTIES)
    write_ply(path, {name: flat[:, k] for k, name in enumerate(GAUSSIAN_PLY_PROPERTIES)})


def load_gaussians_ply(path):
    """scene/gaussian_model.py:408-470: properties are looked up by NAME (scale_*/rot*/normal*/f0* sorted by their numeric suffix)."""
    v, _ = read_ply(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], np.float32)
    family = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    stack = lambda ns: np.stack([col(n) for n in ns], axis=1)
    return {"mean": stack(("x", "y", "z")), "rgb": stack(("f_dc_0", "f_dc_1", "f_dc_2")), "opacity": col("opacity")[:, None], "scale": stack(family("scale_")),
            "rotation": stack(family("rot")), "normal": stack(family("normal")), "roughness": col("roughness")[:, None], "f0": stack(family("f0"))}


def read_init_cloud(path):
    """utils/ply_utils.py:5-13: points [N,3], colours [N,3] (uchar colours -> float / 255)."""
    v, _ = read_ply(path)
    points = np.vstack([v["x"], v["y"], v["z"]]).T
    colors = np.vstack([v["red"], v["green"], v["blue"]]).T
    if colors.dtype == np.uint8:
        colors = colors.astype(np.float32) / 255.0
    return points, colors


def save_init_cloud(path, points, colors):
    """utils/ply_utils.py:16-34: ASCII, six float properties."""
    p, c = np.asarray(points, np.float32), np.asarray(colors, np.float32)
    write_ply(path, {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "red": c[:, 0], "green": c[:, 1], "blue": c[:, 2]}, text=True)


def fov2focal(fov, pixels):  # utils/graphics_utils.py
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def read_transforms(path, width, height, max_images=None):
    """dataset/blender_dataset.py:30-75 without the image buffers: list of dict(file_path, R, T, FovX, FovY, c2w)."""
    with open(path) as f:
        contents = json.load(f)
    frames = sorted(contents["frames"], key=lambda x: x["file_path"])
    if max_images is not None:
        frames = frames[:max_images]
    assert len(frames) != 0, "Dataset is empty"
    fovx = contents["camera_angle_x"]
    fovy = focal2fov(fov2focal(fovx, width), height)
    out = []
    for fr in frames:
        c2w = np.array(fr["transform_matrix"], np.float64)
        c2w[:3, 1:3] *= -1  # Blender (Y up, Z back) -> COLMAP (Y down, Z forward)
        w2c = np.linalg.inv(c2w)
        out.append(dict(file_path=fr["file_path"], R=np.transpose(w2c[:3, :3]), T=w2c[:3, 3], FovX=fovx, FovY=fovy, c2w=c2w))
    return out


def save_cfg(path, cfg):
    """train.py:68-69: json.dump(vars(cfg))."""
    with open(path, "w") as f:
        json.dump(dict(vars(cfg)) if not isinstance(cfg, dict) else cfg, f, indent=2, default=str)


def load_cfg(path):
    with open(path) as f:
        return json.load(f)
