import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import net_select, Singular
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
name=sys.argv[2] if len(sys.argv)>2 else 'SphereNet-ASoftmax'
lr=float(sys.argv[3]) if len(sys.argv)>3 else 0.1
dev='cuda'
g = torch.Generator().manual_seed(0)
images = (torch.rand(B,112,112,3, generator=g)*2-1).to(dev)
labels = torch.randint(0,10575,(B,),generator=torch.Generator().manual_seed(1),dtype=torch.int32).to(dev)
net = net_select(name,'NCHW',5e-4); net.seed=2
step, losses, names, others = Singular(net, lr, 'Momentum')({'images':images,'labels':labels,'num_classes':10575,'num_examples':1})
for i in range(16):
    step(); torch.cuda.synchronize()
    gn = float(net.grads[:net.arena_size].norm()); 
    print(i, [round(float(l),5) for l in losses], 'gnorm %.4g' % gn, 'emb absmax %.4g' % float(net.emb.abs().max()), 'cls gnorm %.4g' % float(net.view('classifier/fc_classifier/weights', net.grads).norm()),
          'wn min %.3g' % (float(net.wn[:10575].min()) if net.head=='asoftmax' else 0))
