"""Build libunet_hip.so (gfx950) in-tree with hipcc.  `python -m unet_amd.build [--force] [--verbose]`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
REPO = ROOT.parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libunet_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", str(REPO / "include"), "-I", str(CSRC),
         "-Wno-unused-value"] + os.environ.get("UNET_EXTRA_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def source_hash() -> str:
    """content hash of the kernel sources + ABI header: names the code a profile / counter file was taken on"""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [REPO / "include" / "unet_hip.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:12]


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [REPO / "include" / "unet_hip.h"]
    srcs = _sources()
    objs = [LIBDIR / (s.stem + ".o") for s in srcs]

    def compile_one(so):
        s, o = so
        if not force and not _stale(o, [s, *headers]):
            return None
        cmd = [HIPCC, *FLAGS, "-c", str(s), "-o", str(o)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s.name}:\n{r.stderr[-8000:]}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        logs = list(ex.map(compile_one, zip(srcs, objs)))
    if verbose:
        (LIBDIR / "resource_usage.log").write_text("\n".join(l for l in logs if l))
    if force or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    build_host_codecs(force)
    return LIB


def build_host_codecs(force: bool = False) -> Path:
    """libunet_tiff.so: the TIFF strip decoders (csrc/tiff_codecs.hip is plain C++) linked by g++ alone -- unet_amd/tiffio.py loads it without
    the HIP runtime (tile preparation on a box without a GPU stack)."""
    # csrc/host/tiff_jpeg.cpp (JPEG-in-TIFF, include/unet_tiff.h) exists in this library only: it is not device-path code and stays out of
    # libunet_hip.so and of source_hash()
    src, jpg, out = CSRC / "tiff_codecs.hip", CSRC / "host" / "tiff_jpeg.cpp", LIBDIR / "libunet_tiff.so"
    LIBDIR.mkdir(exist_ok=True)
    if force or _stale(out, [src, jpg, REPO / "include" / "unet_hip.h", REPO / "include" / "unet_tiff.h"]):
        r = subprocess.run([os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-shared", "-fPIC", "-x", "c++", str(src), str(jpg),
                            "-I", str(REPO / "include"), "-o", str(out)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed for {src.name} / {jpg.name}:\n{r.stderr[-4000:]}")
    return out


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(source_hash())
        sys.exit(0)
    p = build_lib(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(p)
