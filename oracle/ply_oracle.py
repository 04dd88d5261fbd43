"""CPU restatement of the reference's .ply loader (apps/gsrast/SplatData.cpp:28-66, 114-156).
TEST INFRASTRUCTURE, same rules as gsr_oracle.cpp; parity unpinned by the reference (no tests,
no sample .ply ship with it)."""
import numpy as np

F = np.float32


def load(path: str, sh_layout: str = "file") -> dict:
    """sh_layout "coefficient_major": f_rest (channel-major in the file: R1..15, G1..15, B1..15) transposed to
    [15][3] behind the DC triple, the [N][16][3] layout the upstream semantics profile reads (oracle/inria_np.py)."""
    with open(path, "rb") as f:
        lines = [f.readline() for _ in range(3)]                       # three getline calls
        n = int(lines[2].split()[2])                                   # ss >> dummy >> dummy >> numSplats
        while True:
            line = f.readline()
            if not line:
                raise ValueError("no end_header")
            if line.rstrip(b"\n") == b"end_header":
                break
        raw = np.frombuffer(f.read(n * 248), dtype="<f4")
    if raw.size < n * 62:
        raise ValueError("short file")                                  # reader.eof() -> invalid
    r = raw.reshape(n, 62).astype(F)
    means = np.ones((n, 4), F)
    means[:, :3] = r[:, 0:3]
    scales = np.empty((n, 4), F)
    scales[:, :3] = np.exp(r[:, 55:58], dtype=F)
    scales[:, 3] = np.exp(F(1.0), dtype=F)
    q = r[:, 58:62]
    d = (q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + (q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    rot = q * (F(1.0) / np.sqrt(d))[:, None]
    opac = F(1.0) / (F(1.0) + np.exp(-r[:, 54], dtype=F))
    shs = np.ascontiguousarray(r[:, 6:54])
    if sh_layout == "coefficient_major":
        shs = np.concatenate([shs[:, :3], shs[:, 3:].reshape(n, 3, 15).transpose(0, 2, 1).reshape(n, 45)], axis=1)
        shs = np.ascontiguousarray(shs)
    else:
        assert sh_layout == "file"
    return {"means3D": means, "scales": scales, "rotations": rot.astype(F), "opacities": opac.astype(F),
            "shs": shs, "bbox_min": r[:, 0:3].min(0), "bbox_max": r[:, 0:3].max(0),
            "center": r[:, 0:3].astype(np.float64).mean(0).astype(F)}
