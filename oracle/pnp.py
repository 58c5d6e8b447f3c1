"""CPU checks for the PnP tail (TEST INFRASTRUCTURE ONLY).

cv2.solvePnPRansac (utils/pnp_util.py:46-56 of the reference) is not available to pin, so this is not a mirror of the
device kernel but independent checks of what it must deliver: numpy reprojection / inlier recount for any pose, and a
scipy Levenberg-Marquardt (`least_squares`, method "lm") refinement of a pose on a fixed inlier set -- the minimiser the
device's own LM loop has to reach."""
import numpy as np
from scipy.optimize import least_squares


def rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def rotvec(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-12:
        return np.zeros(3)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return w * th


def project(R, t, X, cam):
    fx, fy, cx, cy = cam
    xc = X.astype(np.float64) @ np.asarray(R, np.float64).T + np.asarray(t, np.float64).reshape(1, 3)
    z = xc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = np.stack([fx * xc[:, 0] / z + cx, fy * xc[:, 1] / z + cy], 1)
    return uv, z


def inlier_mask(R, t, X, uv, cam, thresh):
    p, z = project(R, t, X, cam)
    e2 = ((p - uv.astype(np.float64)) ** 2).sum(1)
    return (z > 1e-9) & (e2 <= thresh * thresh)


def refine_lm(R0, t0, X, uv, cam):
    """scipy LM from (R0, t0) on the reprojection error in pixels -> (R, t, rms)."""
    def resid(p):
        q, _ = project(rodrigues(p[:3]), p[3:], X, cam)
        return (q - uv.astype(np.float64)).ravel()
    sol = least_squares(resid, np.concatenate([rotvec(np.asarray(R0, np.float64)), np.asarray(t0, np.float64).ravel()]), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return rodrigues(sol.x[:3]), sol.x[3:], float(np.sqrt((sol.fun ** 2).mean()))
