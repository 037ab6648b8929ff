import torch, ctypes, os, sys
here=os.path.dirname(os.path.abspath(__file__))
for name in ("libprobe.so","libprobe5.so"):
    L=ctypes.CDLL(os.path.join(here,name))
    print(name,"devcount",L.probe_devcount())
    x=torch.zeros(1000,device="cuda")
    s=torch.cuda.current_stream().cuda_stream
    L.probe_fill.argtypes=[ctypes.c_void_p,ctypes.c_int64,ctypes.c_float,ctypes.c_void_p]
    r=L.probe_fill(x.data_ptr(),1000,3.0,s)
    torch.cuda.synchronize()
    print(name,"ret",r,x[:4].tolist(),x[-1].item())
os.system("cat /proc/%d/maps | grep -E 'amdhip|hsa-runtime' | awk '{print $6}' | sort -u"%os.getpid())
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
os.system("nproc; free -g | head -2; rocminfo | grep -E 'Compute Unit|Max Clock|gfx' | head; which rocprofv3")
