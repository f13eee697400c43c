cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
{
python -c "
import torch
p=torch.cuda.get_device_properties(0); print([a for a in dir(p) if 'pci' in a], p.pci_bus_id, p.pci_device_id)
import glob; print(glob.glob('/sys/bus/pci/devices/*/hwmon/hwmon*'))
" 2>&1 | tail -3
python tools/power_probe.py 2>&1 | grep -v "^RCCL\|amdgpu.ids"
SS_LIB_PATH=$PWD/mfcc-rust_amd/lib/libspeechsauce_amd_lab.so SS_WAVES=8 python tools/power_probe.py 2>&1 | grep -v "^RCCL\|amdgpu.ids"
rocm-smi --showpower --showclocks 2>&1 | head -40
} | tee gpurun_out/r04/power_probe.txt
