import sys, contextlib, io, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from visual_odometry_amd import _lib, harness, synth
_lib.require_device(0); torch.cuda.set_device(0)
n=int(sys.argv[1]) if len(sys.argv)>1 else 400
seq = harness.Corridor(n, synth.K_MALAGA, synth.IMAGE_SIZE["malaga"], seed=11, per_frame=900)
with contextlib.redirect_stdout(io.StringIO()):
    t=time.time(); res = harness.run_tracking(seq, min_tracked=int(sys.argv[2]) if len(sys.argv)>2 else 200, new_landmarks="triangulate", track_length=int(sys.argv[3]) if len(sys.argv)>3 else 15); dt=time.time()-t
err=[np.abs(r["T"][:3,3]-np.linalg.inv(seq.world_to_cam(r["frame"]))[:3,3]).max() for r in res]
rot=[np.abs(r["T"][:3,:3]-np.linalg.inv(seq.world_to_cam(r["frame"]))[:3,:3]).max() for r in res]
ho=[r for r in res if r["handover"]]
print(n, dt, "handovers", len(ho), "new lm", [r.get("n_new_landmarks") for r in ho][:12], "max err", max(err), "final", err[-1], "max rot", max(rot), "min inl", min(r["n_inliers"] for r in res))
print([round(e,3) for e in err[::max(1,n//20)]])
