import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cfgs, dynearthsol_amd as des
from dynearthsol_amd.decomp import DeviceGroup
try:
    g = DeviceGroup(des.Host(cfg_text=cfgs.make(**cfgs.EP), overrides="control.has_PT = yes\n", ndims=2), 2)
    print("created; dt", g.init_from_host()); print([s.n_pt_iterations for s in g.step(3)]); g.close()
except Exception as e:
    print("ERROR:", e)
