#!/bin/bash
# tools/after_exit_probe.sh -- is a process slowed down for a while by the memory its PREDECESSOR on the GPU released at exit (the driver
# wipes released VRAM)?  A process that touches 48 GiB exits; straight after it tools/warm_drift_probe.py (basis pass, chunks of 100
# launches against time since its first launch) runs.  Then the same after a 5 s pause.
python3 - <<'EOF'
import torch
x = [torch.ones(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(48)]
torch.cuda.synchronize()
print("predecessor: 48 GiB touched, exiting")
EOF
DRIFT_LEG=M1 python3 tools/warm_drift_probe.py 2>&1 | grep -v amdgpu | awk 'NR<=14 || NR%4==0'
echo "--- the same after 5 s of rest"
sleep 5
DRIFT_LEG=M1 python3 tools/warm_drift_probe.py 2>&1 | grep -v amdgpu | awk 'NR<=8 || NR%6==0'
