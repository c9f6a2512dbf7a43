import os, sys, runpy
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
n_streams = int(sys.argv[2])
import torch
streams = [torch.cuda.Stream() for _ in range(n_streams)]
x = torch.zeros(1024, device="cuda")
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
sys.argv = ["one_pose_tail.py", "1", "4096"]
runpy.run_path("tools/one_pose_tail.py", run_name="__main__")
