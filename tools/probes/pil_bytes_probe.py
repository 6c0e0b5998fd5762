import time, numpy as np, torch
from PIL import Image
def t(f,n=200):
    f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
for (h,w) in [(800,800),(270,480),(700,933),(1080,1920)]:
    a=np.random.default_rng(0).integers(0,255,(h,w,3),dtype=np.uint8)
    im=Image.fromarray(a)
    pin3=torch.empty(h*w*3,dtype=torch.uint8).pin_memory(); pin4=torch.empty(h*w*4,dtype=torch.uint8).pin_memory()
    n3,n4=pin3.numpy(),pin4.numpy()
    def enc_direct(mode_raw, dst):
        e=Image._getencoder(im.mode,"raw",mode_raw); e.setimage(im.im,(0,0)+im.size)
        off=0
        while True:
            l,s,d=e.encode(1<<22)
            dst[off:off+len(d)]=np.frombuffer(d,np.uint8); off+=len(d)
            if s: break
        return off
    print(h,w,'tobytes RGB %.3f'%t(lambda: im.tobytes()),'| +copy to pinned %.3f'%t(lambda: n3.__setitem__(slice(None),np.frombuffer(im.tobytes(),np.uint8))),
          '| tobytes RGBX %.3f'%t(lambda: im.tobytes("raw","RGBX")),'| +copy %.3f'%t(lambda: n4.__setitem__(slice(None),np.frombuffer(im.tobytes("raw","RGBX"),np.uint8))),
          '| encoder->pinned RGB %.3f'%t(lambda: enc_direct("RGB",n3)),'| encoder->pinned RGBX %.3f'%t(lambda: enc_direct("RGBX",n4)))
