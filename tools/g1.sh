cd $GRAFT_REPO_ROOT
python -c "
import torch
from pda_amd import ops
print(ops.measured_peaks())"
export ONLY_ORDER=1
timeout 300 python tools/time_v4.py c1 47890 1 v3,v4 2>&1 | tail -2
