import sys, os
sys.path.insert(0, os.getcwd())
import torch
from stil_tta_amd import STiLModel
from stil_tta_amd.driver import train_step, synthetic_batch
from stil_tta_amd.flat import StilAdam
fl=[8]*16+[1]*48
torch.manual_seed(0)
m=STiLModel(dict(field_lengths=fl,num_classes=286,img_size=224,batch_size=256,start_epoch=35)); m.setup_device("cuda"); m.train(); m.current_epoch=36
m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(286,128)).cuda())
opt=StilAdam(m.flat,lr=1e-4)
b=synthetic_batch(fl,286,256,224,device="cuda")
for i in range(4):
    train_step(m,opt,b)
torch.cuda.synchronize()
st=torch.cuda.memory_stats()
for k in ("allocated_bytes.all.peak","reserved_bytes.all.peak","reserved_bytes.large_pool.peak","inactive_split_bytes.all.peak","active_bytes.all.peak","requested_bytes.all.peak","num_alloc_retries","segment.all.peak"):
    print(k, st.get(k))
print(torch.cuda.memory_summary()[:3000])
