"""Builds libdbw_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain C-ABI shared object
(include/dbw_hip.h); the Python host binds it with ctypes.  Flags matter for parity:
  -ffp-contract=off        one rounding per fp32 op, bit-exact with the CPU oracle's rasteriser arithmetic
  -munsafe-fp-atomics      hardware global/LDS float atomics instead of CAS loops
IEEE fp32 divide/sqrt is hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['util.hip', 'raster.hip', 'project_clip.hip', 'shade_blend.hip', 'render_fused.hip', 'texture.hip', 'model_ops.hip', 'train_step.hip', 'lpips_head.hip']
OUT = os.path.join(HERE, 'dbw_amd', 'libdbw_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-munsafe-fp-atomics',
         '-fno-gpu-flush-denormals-to-zero', '-Wall', '-Wno-unused-function']


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'dbw_hip.h'), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [hipcc] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {s}')
        if verbose and out:
            print(out.decode())
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT])
    return OUT


def build_device_checks(force=False):
    """tests/_build/libdbw_device_checks.so from tests/device_checks.hip: the DEVICE build of the shared host+device headers as a
    checker-side library (same flags as the product), for the GPU tests that hold that arithmetic to the reference's golden vectors."""
    src = os.path.join(HERE, '..', 'tests', 'device_checks.hip')
    out_dir = os.path.join(HERE, '..', 'tests', '_build')
    out = os.path.join(out_dir, 'libdbw_device_checks.so')
    deps = [src] + [os.path.join(CSRC, f) for f in ('dbw_common.h', 'raster_math.h', 'model_math.h', 'rng_math.h')]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps if os.path.exists(d)):
        return out
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.check_call([hipcc] + FLAGS + ['-shared', src, '-o', out])
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, extra_flags=[a for a in sys.argv[1:] if a.startswith('-D')]))
