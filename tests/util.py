"""Helpers shared by the GPU parity tests (HIP path vs torch-CPU oracle ops)."""
import torch

from unet_amd import ops
from unet_amd.ops import TS


def to_ts(x_nchw: torch.Tensor, cs=None, co=0, device="cuda") -> TS:
    """NCHW cpu tensor -> channel slice [co, co+C) of a fresh NHWC device buffer with channel stride cs.
    The rest of the buffer is filled with a sentinel so that out-of-slice reads/writes are caught."""
    N, C, H, W = x_nchw.shape
    cs = ops.rup4(C) if cs is None else cs
    buf = torch.full((N, H, W, cs), 7.25, dtype=torch.float32)
    buf[..., co:co + C] = x_nchw.permute(0, 2, 3, 1)
    if cs > co + C and ops.rup4(C) != C:
        buf[..., co + C:co + ops.rup4(C)] = 0.0   # padded lanes of the slice must be zero
    return TS(buf.to(device).contiguous(), co, C)


def empty_ts(N, H, W, C, cs=None, co=0, device="cuda", fill=7.25) -> TS:
    cs = ops.rup4(C) if cs is None else cs
    buf = torch.full((N, H, W, cs), fill, dtype=torch.float32, device=device)
    return TS(buf, co, C)


def from_ts(t: TS) -> torch.Tensor:
    return t.view().permute(0, 3, 1, 2).contiguous().cpu()


def outside_untouched(t: TS, fill=7.25) -> bool:
    b = t.buf.cpu()
    mask = torch.ones(b.shape[-1], dtype=torch.bool)
    mask[t.co:t.co + t.C] = False
    return bool((b[..., mask] == fill).all())


def assert_close(got: torch.Tensor, ref: torch.Tensor, rtol=1e-4, atol=1e-5, what=""):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= rtol * scale + atol, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"
