cd /root/repo
python - <<'PY'
import os,sys,subprocess,tempfile,re
sys.path.insert(0,'svt-av1-mod-by-patman_amd'); sys.path.insert(0,'tools')
import numpy as np
from svtav1_hip import frames
W,H,N=1920,1080,33
tmp=tempfile.mkdtemp(dir='/tmp'); path=tmp+'/c.yuv'
with open(path,'wb') as f:
    for y in frames.synthetic_clip(W,H,N,seed=7):
        f.write(y.tobytes()); f.write(np.full((H//2)*(W//2)*2,128,np.uint8).tobytes())
env=dict(os.environ, SVTAV1_E2E_SIMD='2', SVTAV1_HIP_LIB=os.environ.get("HOOKTIME_LIB", os.getcwd()+"/svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip.so"), SVTAV1_HIP_ONLY='__none__')
for h in ('PA','ME','TF','TPL','DLF','CDEF','LR'): env['SVTAV1_HIP_TIERB_'+h]='1'
r=subprocess.run(['oracle/_ref/e2e/SvtAv1EncApp','-i',path,'-w',str(W),'-h',str(H),'--fps','30','-n',str(N),'--preset','8','--lp','16','--asm','hip','-b',tmp+'/o.ivf'],env=env,stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
print('\n'.join(l for l in r.stdout.split('\n') if 'svt_hip_bind' in l or 'Average Speed' in l))
PY
