"""CPU restatement of the reference's .ply loader (apps/gsrast/SplatData.cpp:28-66, 114-156).
TEST INFRASTRUCTURE, same rules as gsr_oracle.cpp; parity unpinned by the reference (no tests,
no sample .ply ship with it)."""
import numpy as np

F = np.float32


def load(path: str) -> dict:
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
    return {"means3D": means, "scales": scales, "rotations": rot.astype(F), "opacities": opac.astype(F),
            "shs": np.ascontiguousarray(r[:, 6:54]), "bbox_min": r[:, 0:3].min(0), "bbox_max": r[:, 0:3].max(0),
            "center": r[:, 0:3].astype(np.float64).mean(0).astype(F)}
