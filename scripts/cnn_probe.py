"""ResNet-50 extractor throughput: dtype x batch x MIOpen find mode (dev aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_engine_amd.descriptors import CNNDescriptor
dev = torch.device("cuda", 0)
for fold in (False, True):
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        cnn = CNNDescriptor(out_dim=512, dtype=dt, fold_bn=fold)
        for B in (256,):
            imgs = torch.randint(0, 256, (B, 224, 224, 3), device=dev, dtype=torch.uint8)
            for _ in range(3): cnn.extract_features_tensor(imgs)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): cnn.extract_features_tensor(imgs)
            torch.cuda.synchronize(); el = (time.perf_counter() - t) / 5
            print(f"fold_bn={fold} {name} B={B}: {B/el:8.0f} img/s", flush=True)
