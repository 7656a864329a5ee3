import sys, json
sys.path.insert(0,'/root/repo')
import torch, bench
from ssv_amd.utils import augmentations
dev=torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(420)
source = torch.randint(0, 256, (32, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
ids = torch.arange(32, device=dev, dtype=torch.int64)
cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench.AUG_CFG.items()}
tf = augmentations.get_transform(cfg)
views = tf.apply(source, ids, tf.draw(source, ids, 0))
print(json.dumps(bench.update_check(dev, sys.argv[1], views[0], views[1]), indent=1))
