import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import dynearthsol_amd as des
from oracle_binding import OracleEngine
import cfgs
def run(name, kw, nsteps, chunks):
    host = des.Host(cfg_text=cfgs.make(**kw))
    dev = des.DeviceEngine(host); ora = OracleEngine(host)
    dtd = dev.init_from_host(host); dto = ora.init_from_host(host)
    print(name, 'nn,ne', host.nnode, host.nelem, 'dt', dtd, dto, dtd==dto)
    for f in ("VOLUME","VOLUME_N","MASS","TMASS","VEL"):
        a,b = dev.download(f), ora.download(f)
        print('  init', f, np.array_equal(a,b), np.abs(a-b).max())
    for c in range(chunks):
        sd = dev.step(nsteps); so = ora.step(nsteps)
        print('  steps', sd.steps, so.steps, 'dt', sd.dt, so.dt, 'time', sd.time==so.time, 'l2', sd.l2_residual, so.l2_residual, 'msv', sd.max_surf_vel, so.max_surf_vel)
        for f in ("COORD","VEL","FORCE","TEMPERATURE","STRESS","STRAIN","STRAIN_RATE","PLSTRAIN","DELTA_PLSTRAIN","VISCOSITY","VOLUME","VOLUME_OLD","VOLUME_N","MASS","TMASS","DPRESSURE","DH","DHACC","EDVACC_SURF"):
            a,b = dev.download(f), ora.download(f)
            m = np.abs(b).max()
            print('   %-14s exact=%s  maxrel=%.3e' % (f, np.array_equal(a,b), (np.abs(a-b).max()/m if m>0 else np.abs(a-b).max())))
        print('   nyield', (ora.download("DELTA_PLSTRAIN")>0).sum(), 'nan', dev.check_nan())
run('EP', cfgs.EP, 25, 2)
run('EVP', cfgs.EVP, 25, 2)
run('YIELD', cfgs.YIELD, 50, 2)
run('EVP2mat', dict(cfgs.EVP, nmat=2), 25, 1)
