import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    import nabo_amd
    print("nabo devs", nabo_amd.device_count())
else:
    import nabo_amd
    print("nabo devs", nabo_amd.device_count())
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
import subprocess
print(subprocess.run("grep -i hip /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, stdout=subprocess.PIPE, universal_newlines=True).stdout)
