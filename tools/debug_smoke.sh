cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
python -c "
import __graft_entry__ as g
g.build()
import os
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l][:6])
import torch
print('torch', torch.__file__)
print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa-runtime' in l)))
g.smoke()" 2>&1 | tail -4 | cut -c1-400
ldd dreamwaltz-g_amd/csrc/libdwg_hip.so | grep -i "hip\|hsa"
