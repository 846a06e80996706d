import os, glob
print("affinity:", sorted(os.sched_getaffinity(0)))
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip()[:80])
    except Exception as e:
        print(d, "ERR", e)
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    try:
        print(n, open(n + "/cpulist").read().strip()[:100])
    except Exception as e:
        print(n, "ERR", e)
import torch
print(torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else "no pci id attr")
try:
    import subprocess
    print(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showtoponuma"], capture_output=True, text=True, timeout=20).stdout[-600:])
except Exception as e:
    print("smi ERR", e)
