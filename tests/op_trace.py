"""Which ATen operators a call reaches (test infrastructure): the dispatcher's RecordFunction hooks see every operator, also the
ones a C++ extension operator calls internally (at::linear inside torch.ops.pbllm_native.linear) -- how the GPU tests assert that
the GEMM regime never leaves the hand-written kernels for a library GEMM."""
import torch

LIBRARY_GEMM_OPS = {"aten::linear", "aten::mm", "aten::addmm", "aten::matmul", "aten::bmm", "aten::baddbmm", "aten::_scaled_mm"}


def called_ops(fn) -> set:
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        fn()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    return {e.name for e in prof.events()}
