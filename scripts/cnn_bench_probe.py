"""ResNet-50 extractor throughput: MIOpen find mode (torch.backends.cudnn.benchmark) on/off, fp32 / bf16, batch sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_engine_amd.descriptors import CNNDescriptor
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for dtype in (torch.float32, torch.bfloat16):
        desc = CNNDescriptor(dtype=dtype)
        for bs in (64, 256):
            x = torch.randint(0, 256, (bs, 224, 224, 3), dtype=torch.uint8, device="cuda")
            for _ in range(3): desc.extract_features_tensor(x)
            torch.cuda.synchronize(); t0 = time.time()
            reps = 10
            for _ in range(reps): desc.extract_features_tensor(x)
            torch.cuda.synchronize(); dt = time.time() - t0
            print(f"benchmark={bench} {str(dtype).split('.')[-1]} batch {bs}: {bs * reps / dt:.0f} img/s", flush=True)
