#!/usr/bin/env python3
"""Probe of the split-bf16 dense layer (hnr_linear_s3): error against fp64 beside the fp32-MFMA kernel, and time per launch."""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridneuralrendering_amd.linear import PackedLinear, SplitLinear

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4000000)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)

def check(M, K, lda, side=False):
    A = torch.randn((M, lda), device=dev) * (torch.rand((M, 1), device=dev) * 4)
    W = torch.randn((256, K), device=dev) / K ** 0.5
    b = torch.randn((256,), device=dev)
    ref = A[:, :K].double() @ W.double().t() + b.double()
    R = ridx = None
    if side:
        R = torch.randn((1000, 256), device=dev)
        ridx = torch.randint(0, 1000, (M,), device=dev, dtype=torch.int32)
        ref = ref + R.double()[ridx.long()]
    ref = torch.where(ref > 0, ref, ref * 0.01)
    A2 = A.clone()
    if lda > K:
        A2[:, K:] = float("nan")          # padding columns must never enter a product
    s3, f32 = SplitLinear(W, b), PackedLinear(W, b)
    if side:
        o3 = s3.gather_add(A2, R, ridx, act=True, K=K); o1 = f32.gather_add(A2, R, ridx, act=True, K=K)
    else:
        o3 = s3(A2, act=True, K=K); o1 = f32(A2, act=True, K=K)
    scale = (A[:, :K].abs().double() @ W.abs().double().t()).clamp_min(1e-30)
    e3 = ((o3.double() - ref).abs() / scale).max().item()
    e1 = ((o1.double() - ref).abs() / scale).max().item()
    print("M=%d K=%d lda=%d side=%d: max |err| / sum|a||w|  split-bf16 %.3e   fp32-MFMA %.3e   max abs diff between them %.3e"
          % (M, K, lda, side, e3, e1, (o3 - o1).abs().max().item()), flush=True)

for M, K, lda, side in [] if os.environ.get("HNR_S3_DBG") else [(256, 256, 256, False), (1000, 256, 256, False), (777, 263, 264, False), (5000, 60, 64, True), (70001, 256, 288, False)]:
    check(M, K, lda, side)

M = a.rows
A = torch.randn((M, 256), device=dev)
W = torch.randn((256, 256), device=dev) / 16
b = torch.randn((256,), device=dev)
out = torch.empty((M, 256), device=dev)
if os.environ.get("HNR_S3_DBG"):
    print("ABLATION HNR_S3_DBG=%s (results are wrong on purpose)" % os.environ["HNR_S3_DBG"])
for name, layer in (("split-bf16", SplitLinear(W, b)), ("fp32-MFMA", PackedLinear(W, b)))[:1 if os.environ.get("HNR_S3_DBG") else 2]:
    layer(A, out=out, act=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        layer(A, out=out, act=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    if os.environ.get("HNR_S3_DBG") == "256":
        ref = torch.nn.functional.leaky_relu(A[:200000].double() @ W.double().t() + b.double(), 0.01)
        mag = A[:200000].abs().double() @ W.abs().double().t() + b.abs().double()
        print("3-term variant: max |err| / sum|a||w| = %.3e" % ((out[:200000].double() - ref).abs() / mag).max().item())
    print("%s: M=%d 256x256: %.3f ms  = %.1f fp32-equivalent TFLOP/s, %.2f TB/s of A+C" % (name, M, ms, 2.0 * M * 65536 / ms / 1e9, M * 2048 / ms / 1e9), flush=True)
