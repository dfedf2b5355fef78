import sys, time, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'mask-yolo_amd')]
import torch
from myolo.config import make_config, ShapesConfig
from myolo.model import MaskYOLO
from myolo.shapes import make_shapes_samples
from myolo.myolo_utils import BatchGenerator
cfg = make_config(ShapesConfig, IMAGE_SHAPE=[224, 224, 3], ALPHA=1.0, BATCH_SIZE=32)
net = MaskYOLO(mode="training", config=cfg, seed=0).net
dbs=[]
for k in range(2):
    batch,_=BatchGenerator(make_shapes_samples(32,cfg,start_index=32*k),cfg,'training',shuffle=False,norm=True)[0]
    dbs.append(net.to_device_batch(batch))
for i in range(8): net.train_step(dbs[i%2],1e-3)
def run(s,steps=20):
    torch.cuda.synchronize(); net.host_wait_s=0.0; t0=time.perf_counter()
    for i in range(steps):
        if s: time.sleep(s)
        net.train_step(dbs[i%2],1e-3)
    torch.cuda.synchronize()
    return 1e3*(time.perf_counter()-t0)/steps, 1e3*net.host_wait_s/steps
for s in (0,0.002,0.004,0.008,0.012,0.016,0.020,0.030):
    a,b=run(s); print("sleep %4.0f ms per step: step %.2f ms, host blocked on n_pos %.2f ms" % (1e3*s,a,b))
