import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
print("torch", torch.__version__, "cuda avail", torch.cuda.is_available(), torch.cuda.device_count())
print(torch.cuda.get_device_name(0))
import ctypes
x = torch.zeros(4, device="cuda"); print(x)
os.environ["FRLW_DEBUG"]="1"
from frlw_evd_amd import _lib
l = _lib.load()
print(l.frlw_version())
os.system("cat /proc/%d/maps | grep -i amdhip | awk '{print $6}' | sort -u" % os.getpid())
import numpy as np
from frlw_evd_amd import event_representation as er, synth
ev = synth.synth_events(1, 1000, 12, 8, 1000)
e = torch.from_numpy(synth.to_xytp_f64(ev, ev["t"]/1000.0)).cuda()
try:
    out, dt = er.generate_eventframe(e, (8, 12))
    print(out.sum().item(), dt)
except Exception as ex:
    print("EXC", repr(ex))
t0=time.time()
ev = synth.synth_events(1003, 10_000_000, 1280, 720, 80_000)
print("synth 10M", time.time()-t0)
dat = torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()
st = torch.full((720,1280, 2, 8), -6000.0, device="cuda")
t0=time.time()
u8, _ = er.encode_taf_dat(dat, (720,1280), st, 0, 10_000, 8, 8)
torch.cuda.synchronize(); print("encode", time.time()-t0, st.mean().item(), u8.float().mean().item())
for i in range(3):
    st.fill_(-6000.0); torch.cuda.synchronize(); t0=time.time()
    u8, _ = er.encode_taf_dat(dat, (720,1280), st, 0, 10_000, 8, 8, check=False)
    torch.cuda.synchronize(); print("encode", time.time()-t0)
