cd $GRAFT_REPO_ROOT
nproc; free -g | head -2
timeout 300 python - <<'PY'
import time, torch, sys, warnings
sys.path.insert(0,'.')
t0=time.time()
import fcd_gan_pytorch_amd as fcd
from fcd_gan_pytorch_amd.synthetic import synthetic_tiles
dev='cuda'
def T(msg):
    torch.cuda.synchronize(); print('%7.2fs %s'%(time.time()-t0,msg), flush=True)
C,H,W,N=13,256,256,2
torch.manual_seed(0)
netS=fcd.Module.Segmentor(C,bilinear=True).to(dev).train()
netD=fcd.Module.Discriminator_SRGAN_simple(C).to(dev).train()
netG=fcd.Module.Generator(C).to(dev).eval()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    crit=fcd.Loss.CGeneratorLoss(channel=C,perception_layer=1,perception_perBand=True).to(dev)
T('built')
x,y,region=(t.to(dev) for t in synthetic_tiles(1234,N,C,H,W))
for it in range(2):
    cmap=netS(x,y); T('S fwd')
    cmap.mean().backward(); T('S bwd')
    o=netD(x*(1-cmap.detach()),y); T('D fwd')
    o.mean().backward(); T('D bwd')
    with torch.no_grad(): yf=netG(x)
    T('G fwd')
    cm=cmap.detach().requires_grad_(True)
    g,s,p=crit(y,yf,cm); T('crit fwd')
    (g+0.1*p+0*s).backward(); T('crit bwd')
PY
