"""Gaussian `.ply` interchange — numpy only, no `plyfile` dependency.

Writes / reads the binary-little-endian vertex table that 3DGS viewers expect and that the reference
produces in ``ggrt/model/pixelsplat/ply_export.py:12-23`` (attribute list) and ``:26-92`` (scene
normalisation + write):

    x y z  nx ny nz  f_dc_0..2  [f_rest_*]  opacity  scale_0..2 (log)  rot_0..3 (w,x,y,z)

``export_ply`` follows the reference's semantics: shift by the per-axis median, divide by the largest
per-axis 95 % quantile of |means| (scales too), rotate into the viewer frame
(45° about −z ∘ axis swizzle ∘ w2c rotation), compose that rotation onto every Gaussian's quaternion,
export the SH DC band only, store log-scales, opacities as given — in the reference's order of float32 / float64
operations and with its quaternion sign convention, so that the vertex table is the reference's byte for byte
(``tests/golden/ply_export_scene.npz`` holds the table the reference's own ``export_ply`` handed to ``plyfile`` for a
seeded scene, ``tests/test_ply_io.py``).  The reference needs ``plyfile`` and ``scipy``; neither is needed here: the two
scipy conversions are restated in numpy (``quat_xyzw_to_matrix``, ``matrix_to_quat_xyzw_markley``), the container
format (header text + little-endian rows) is the PLY 1.0 binary layout.

``import_ply`` / ``load_gaussians`` read such a file back into rasterizer-boundary tensors
(``means3D, scales, rotations, opacities, shs[P,1+rest,3]``) so a scene can travel as a golden fixture.
"""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import torch


def construct_list_of_attributes(num_rest: int) -> list[str]:
    """Column order of the vertex element (reference ``ply_export.py:12-23``)."""
    cols = ["x", "y", "z", "nx", "ny", "nz"]
    cols += [f"f_dc_{i}" for i in range(3)]
    cols += [f"f_rest_{i}" for i in range(num_rest)]
    cols += ["opacity"]
    cols += [f"scale_{i}" for i in range(3)]
    cols += [f"rot_{i}" for i in range(4)]
    return cols


def _header(num_vertices: int, columns: list[str]) -> bytes:
    lines = ["ply", "format binary_little_endian 1.0", f"element vertex {num_vertices}"]
    lines += [f"property float {c}" for c in columns]
    lines += ["end_header", ""]
    return "\n".join(lines).encode("ascii")


def write_vertex_table(path, table: np.ndarray, columns: list[str]) -> None:
    """``table`` [N, len(columns)] float32 → binary little-endian PLY."""
    table = np.ascontiguousarray(table, dtype="<f4")
    if table.ndim != 2 or table.shape[1] != len(columns):
        raise ValueError(f"table shape {table.shape} does not match {len(columns)} columns")
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    with open(path, "wb") as f:
        f.write(_header(table.shape[0], columns))
        f.write(table.tobytes())


def read_vertex_table(path) -> tuple[np.ndarray, list[str]]:
    """Inverse of :func:`write_vertex_table` (float properties, binary little-endian, one element)."""
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.find(b"end_header\n")
    if not blob.startswith(b"ply\n") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    head = blob[:end].decode("ascii").split("\n")
    if "format binary_little_endian 1.0" not in head:
        raise ValueError(f"{path}: only binary_little_endian 1.0 is supported")
    n, columns, elements = None, [], 0
    for line in head:
        tok = line.split()
        if tok[:1] == ["element"]:
            elements += 1
            if tok[1] != "vertex" or elements > 1:
                raise ValueError(f"{path}: expected a single 'vertex' element")
            n = int(tok[2])
        elif tok[:1] == ["property"]:
            if tok[1] not in ("float", "float32"):
                raise ValueError(f"{path}: property {tok[-1]} has unsupported type {tok[1]}")
            columns.append(tok[2])
    if n is None:
        raise ValueError(f"{path}: no vertex element")
    body = blob[end + len(b"end_header\n"):]
    need = n * len(columns) * 4
    if len(body) < need:
        raise ValueError(f"{path}: truncated ({len(body)} of {need} payload bytes)")
    table = np.frombuffer(body[:need], dtype="<f4").reshape(n, len(columns)).copy()
    return table, columns


def quat_wxyz_to_matrix(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return m.reshape(*q.shape[:-1], 3, 3)


def matrix_to_quat_wxyz(m: np.ndarray) -> np.ndarray:
    """Rotation matrices [...,3,3] → unit quaternions (w,x,y,z), w ≥ 0; branch on the largest of
    (trace, m00, m11, m22) for numerical safety."""
    q = matrix_to_quat_xyzw_markley(np.asarray(m, dtype=np.float64).reshape(-1, 3, 3), orthogonalize=False)
    q = q[:, [3, 0, 1, 2]].reshape(*np.shape(m)[:-2], 4)
    return np.where(q[..., :1] < 0, -q, q)


def matrix_to_quat_xyzw_markley(m: np.ndarray, orthogonalize: bool = True) -> np.ndarray:
    """[N,3,3] float64 → (x,y,z,w) with the conventions of the function the reference calls at
    ``ply_export.py:69`` (``scipy.spatial.transform.Rotation.from_matrix(...).as_quat()``): a matrix whose Gramian is not
    the identity to 1e-12 (here: every one, the viewer rotation is float32) is first replaced by the nearest orthogonal
    matrix U·Vᵀ of its SVD, then Markley's 2008 rule picks the largest of (m00, m11, m22, trace) and builds the
    quaternion around it; the SIGN is whatever that rule yields (no w ≥ 0 canonicalisation) — q and −q are the same
    rotation, but a byte-identical file needs the same one."""
    m = np.array(m, dtype=np.float64, copy=True)
    if orthogonalize and len(m):
        bad = ~np.all(np.isclose(m @ np.swapaxes(m, 1, 2), np.eye(3), atol=1e-12), axis=(1, 2))
        if bad.any():
            u, _, vt = np.linalg.svd(m[bad])
            m[bad] = u @ vt
    n = len(m)
    dec = np.empty((n, 4))
    dec[:, 0], dec[:, 1], dec[:, 2] = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    dec[:, 3] = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    choice = dec.argmax(axis=1)
    q = np.empty((n, 4))
    idx = np.nonzero(choice != 3)[0]
    i = choice[idx]
    j = (i + 1) % 3
    k = (j + 1) % 3
    q[idx, i] = 1 - dec[idx, 3] + 2 * m[idx, i, i]
    q[idx, j] = m[idx, j, i] + m[idx, i, j]
    q[idx, k] = m[idx, k, i] + m[idx, i, k]
    q[idx, 3] = m[idx, k, j] - m[idx, j, k]
    idx = np.nonzero(choice == 3)[0]
    q[idx, 0] = m[idx, 2, 1] - m[idx, 1, 2]
    q[idx, 1] = m[idx, 0, 2] - m[idx, 2, 0]
    q[idx, 2] = m[idx, 1, 0] - m[idx, 0, 1]
    q[idx, 3] = 1 + dec[idx, 3]
    return q / np.linalg.norm(q, axis=1)[:, None]


def quat_xyzw_to_matrix(q: np.ndarray) -> np.ndarray:
    """(x,y,z,w) [N,4] float64 → [N,3,3]: normalise, then the products in the order the function the reference calls at
    ``ply_export.py:67`` (``Rotation.from_quat(...).as_matrix()``) forms them."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=1)[:, None]
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    m = np.empty((len(q), 3, 3))
    m[:, 0, 0] = x2 - y2 - z2 + w2
    m[:, 1, 0] = 2 * (xy + zw)
    m[:, 2, 0] = 2 * (xz - yw)
    m[:, 0, 1] = 2 * (xy - zw)
    m[:, 1, 1] = -x2 + y2 - z2 + w2
    m[:, 2, 1] = 2 * (yz + xw)
    m[:, 0, 2] = 2 * (xz + yw)
    m[:, 1, 2] = 2 * (yz - xw)
    m[:, 2, 2] = -x2 - y2 + z2 + w2
    return m


def viewer_rotation(extrinsics: torch.Tensor) -> torch.Tensor:
    """3×3 world→viewer rotation of the export (reference ``ply_export.py:43-63``): +Z up swizzle, a −45°
    turn about z for the viewer's start pose, then the camera-to-world rotation undone — formed in float32 in the
    reference's order of products, so the table comes out bit for bit."""
    swizzle = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]], dtype=torch.float32)
    # rotation vector (0, 0, −45°) → quaternion (0, 0, sin(a/2), cos(a/2)) → matrix, in float64, then float32
    a = math.radians(-45.0)
    turn = torch.tensor(quat_xyzw_to_matrix(np.array([[0.0, 0.0, math.sin(a / 2), math.cos(a / 2)]]))[0],
                        dtype=torch.float32)
    return (turn @ swizzle) @ extrinsics[:3, :3].detach().float().cpu().inverse()


def export_ply(extrinsics: torch.Tensor, means: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor,
               harmonics: torch.Tensor, opacities: torch.Tensor, path) -> None:
    """Same arguments as the reference's ``export_ply`` (``ply_export.py:26-34``): ``rotations`` are
    (x,y,z,w) quaternions (scipy order, as the reference feeds ``R.from_quat``), ``harmonics`` is
    ``[G,3,d_sh]`` and only its DC band is written; the file stores (w,x,y,z) with the sign the reference's
    conversion yields.  The vertex table equals the one the reference hands to ``plyfile`` byte for byte
    (``tests/golden/ply_export_scene.npz``, recorded from the reference's own function)."""
    means = means.detach().float().cpu()
    scales = scales.detach().float().cpu()
    means = means - means.median(dim=0).values
    factor = means.abs().quantile(0.95, dim=0).max()
    means, scales = means / factor, scales / factor
    rot = viewer_rotation(extrinsics)
    means = torch.einsum("ij,gj->gi", rot, means)
    local = quat_xyzw_to_matrix(rotations.detach().cpu().numpy())
    q = matrix_to_quat_xyzw_markley(rot.numpy() @ local)
    dc = harmonics.detach().float().cpu()[..., 0]
    table = np.concatenate([means.numpy(), np.zeros_like(means.numpy()), dc.contiguous().numpy(),
                            opacities.detach().float().cpu().numpy()[:, None], scales.log().numpy(),
                            q[:, [3, 0, 1, 2]]], axis=1).astype(np.float32)
    write_vertex_table(path, table, construct_list_of_attributes(0))


def save_gaussians(path, means3D, scales, rotations, opacities, shs) -> None:
    """Lossless dump of rasterizer-boundary tensors (``rotations`` (w,x,y,z), ``shs`` [P,M,3]); f_rest is
    channel-major like 3DGS checkpoints ([P,3,M-1] flattened)."""
    P, M = shs.shape[0], shs.shape[1]
    sh = shs.detach().float().cpu()
    rest = sh[:, 1:].permute(0, 2, 1).reshape(P, 3 * (M - 1))
    table = np.concatenate([means3D.detach().float().cpu().numpy(), np.zeros((P, 3), np.float32), sh[:, 0].numpy(),
                            rest.numpy(), opacities.detach().float().cpu().reshape(P, 1).numpy(),
                            scales.detach().float().cpu().log().numpy(), rotations.detach().float().cpu().numpy()], axis=1)
    write_vertex_table(path, table, construct_list_of_attributes(3 * (M - 1)))


def load_gaussians(path, device="cpu") -> dict:
    """Read a Gaussian `.ply` into the tensors ``GaussianRasterizer`` takes (scales are exponentiated)."""
    table, columns = read_vertex_table(path)
    col = {c: i for i, c in enumerate(columns)}
    for need in ("x", "y", "z", "f_dc_0", "opacity", "scale_0", "rot_0"):
        if need not in col:
            raise ValueError(f"{path}: missing column {need}")
    n_rest = sum(c.startswith("f_rest_") for c in columns)
    if n_rest % 3:
        raise ValueError(f"{path}: {n_rest} f_rest columns is not a multiple of 3")
    t = torch.from_numpy(table)
    pick = lambda names: t[:, [col[n] for n in names]]
    dc = pick([f"f_dc_{i}" for i in range(3)])[:, None, :]
    rest = pick([f"f_rest_{i}" for i in range(n_rest)]).reshape(-1, 3, n_rest // 3).permute(0, 2, 1)
    out = dict(means3D=pick(["x", "y", "z"]), scales=pick([f"scale_{i}" for i in range(3)]).exp(),
               rotations=pick([f"rot_{i}" for i in range(4)]), opacities=pick(["opacity"]),
               shs=torch.cat([dc, rest], 1).contiguous())
    return {k: v.contiguous().to(device) for k, v in out.items()}
