"""Dev probe: what HBM bandwidth do plain torch fill / copy / read-reduce kernels reach on this box?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tools_dev.conv_probe import timeit

for mb in (189, 757):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
    us = timeit(lambda: x.zero_());           print(f"fill   {mb:4d} MB: {us:7.1f} us  {mb / us:5.2f} TB/s (write)")
    us = timeit(lambda: y.copy_(x));          print(f"copy   {mb:4d} MB: {us:7.1f} us  {2 * mb / us:5.2f} TB/s (read+write)")
    us = timeit(lambda: x.sum());             print(f"sum    {mb:4d} MB: {us:7.1f} us  {mb / us:5.2f} TB/s (read)")
    h = x.to(torch.bfloat16)
    us = timeit(lambda: h.float());           print(f"widen  {mb // 2:4d}->{mb} MB: {us:7.1f} us  {1.5 * mb / us:5.2f} TB/s")
