"""CPU: libdbw_hip.so builds for gfx950, loads without a GPU, exports every symbol include/dbw_hip.h declares, and the
ctypes signatures in dbw_amd/_lib.py match the header prototypes argument by argument.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from dbw_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'dbw_hip.h')
CTYPE = {'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double, 'int64_t': ctypes.c_int64, 'size_t': ctypes.c_size_t,
         'dbw_stream_t': ctypes.c_void_p}


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r'\b(int64_t|int|size_t|void \*|void|const char \*|dbw_step_plan \*)\s*(dbw_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        args = ' '.join(args.split())
        types = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    types.append(ctypes.c_void_p)
                else:
                    types.append(CTYPE[a.replace('const ', '').split()[0]])
        protos[name] = (ret.strip(), types)
    return protos


def test_library_builds_and_loads_without_gpu():
    lib = _lib.load()
    assert os.path.exists(_lib.LIB_PATH)
    assert lib.dbw_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define DBW_ABI_VERSION (\d+)', open(HEADER).read()).group(1))
    assert lib.dbw_last_error() is not None


def test_every_declared_symbol_is_exported_with_matching_signature():
    lib = _lib.load()
    protos = parse_header()
    assert len(protos) == 57
    for name, (ret, types) in protos.items():
        assert hasattr(lib, name), f'{name} declared in dbw_hip.h but not exported'
        if name in _lib.SIGNATURES:
            assert _lib.SIGNATURES[name] == types, f'{name}: ctypes signature differs from the header'
    missing = set(_lib.SIGNATURES) - set(protos)
    assert not missing, f'bound but not declared: {missing}'
    RET = {'int': ctypes.c_int, 'size_t': ctypes.c_size_t, 'int64_t': ctypes.c_int64, 'void': None, 'void *': ctypes.c_void_p, 'dbw_step_plan *': ctypes.c_void_p}
    for name, (restype, argtypes) in _lib.OTHER_SIGNATURES.items():
        assert name in protos, f'{name} bound but not declared'
        assert protos[name] == ([k for k, v in RET.items() if k == protos[name][0]][0], argtypes) and RET[protos[name][0]] == restype, name
        fn = getattr(lib, name)
        assert fn.restype == restype and fn.argtypes == argtypes, name
    undeclared_compute = {n for n in protos if n not in _lib.SIGNATURES and n not in _lib.OTHER_SIGNATURES} - {'dbw_abi_version', 'dbw_bin_subcursors', 'dbw_last_error', 'dbw_rasterize_workspace_bytes', 'dbw_rasterize_workspace_bytes_binned', 'dbw_debug_set_flags', 'dbw_debug_set_raster_flags'}
    assert not undeclared_compute, f'declared but not bound: {undeclared_compute}'


def test_bin_cursor_count_comes_from_the_library():
    """The host sizes the cursor array and rounds the record capacity of the texture bins with the library's DBW_BIN_SUBCURSORS
    (dbw_bin_subcursors), not with a constant of its own."""
    import re
    from dbw_amd import ops
    lib = _lib.load()
    declared = int(re.search(r'#define DBW_BIN_SUBCURSORS (\d+)', open(HEADER).read()).group(1))
    assert lib.dbw_bin_subcursors() == declared == ops.BIN_SUBCURSORS == ops.bin_subcursors()
    for nbins in (1, 7, 640):
        cap = ops.texbin_capacity(49, 300, 400, 10, nbins)
        assert cap % declared == 0 and cap >= 256


def test_argument_validation_happens_before_any_launch():
    """Null pointers are rejected by the ABI itself (no GPU needed: validation precedes the launch)."""
    lib = _lib.load()
    rc = lib.dbw_rasterize_fwd(0, 0, 0, 0, 1, 0, 8, 8, 2, 0.0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b'null pointer' in lib.dbw_last_error()
    with pytest.raises(RuntimeError, match='null pointer'):
        _lib.call('dbw_tv_l2sq', 0, 1, 4, 4, 0, 1.0, 0, 0, 0)


def test_graft_entry_build_runs_on_the_committed_tree():
    """The driver's "does it build" check: __graft_entry__.build() compiles (or finds up to date) every HIP source and the CPU oracle and
    returns the library; it used to assert a stale ABI literal."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_graft_entry_under_test', os.path.join(ROOT, '__graft_entry__.py'))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    so = ge.build()
    assert os.path.exists(so) and so.endswith('libdbw_hip.so')


def _parse_struct(name):
    """[(field, ctype)] of `typedef struct name { ... } name;` in the header: plain declarations only (scalars, pointers, small arrays)."""
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), src, flags=re.S).group(1)
    base = dict(CTYPE, uint64_t=ctypes.c_uint64, int32_t=ctypes.c_int32)
    fields = []
    for decl in body.split(';'):
        decl = ' '.join(decl.split())
        if not decl:
            continue
        m = re.match(r'(const )?(\w+) (.*)', decl)
        t = base[m.group(2)]
        for item in m.group(3).split(','):
            item = item.strip()
            arr = re.match(r'(\w+)\[(\d+)\]', item)
            if item.startswith('*'):
                fields.append((item.lstrip('* '), ctypes.c_void_p))
            elif arr:
                fields.append((arr.group(1), t * int(arr.group(2))))
            else:
                fields.append((item, t))
    return fields


def test_step_structures_match_the_header_field_by_field():
    """dbw_step_desc / dbw_step_inputs are filled through ctypes.Structure mirrors: same fields, same order, same types, or the library
    reads garbage."""
    for cname, cls in (('dbw_step_desc', _lib.StepDesc), ('dbw_step_inputs', _lib.StepInputs)):
        want = _parse_struct(cname)
        got = list(cls._fields_)
        assert [n for n, _ in want] == [n for n, _ in got], cname
        for (n, a), (_, b) in zip(want, got):
            assert ctypes.sizeof(a) == ctypes.sizeof(b) and (a is b or a._type_ == b._type_ or {a, b} <= {ctypes.c_int, ctypes.c_int32}), (cname, n, a, b)


def test_train_step_rejects_a_bad_descriptor_without_a_gpu():
    lib = _lib.load()
    d = _lib.StepDesc()
    assert lib.dbw_train_step_workspace_bytes(ctypes.byref(d)) == 0 and b'dbw' not in lib.dbw_last_error()[:0]
    assert not lib.dbw_train_step_create(ctypes.byref(d), 0, 0)
    assert lib.dbw_train_step_offset(None, 0) == -1


def test_header_is_plain_c_and_a_c_host_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/dbw_hip.h compiles as C99 with warnings as errors (no C++ in it outside the extern "C" guard), and a
    host written in C links against libdbw_hip.so and calls it -- here the two entry points that need no GPU, plus the address of every
    function the header declares (an undefined symbol fails the link)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    _lib.load()
    names = sorted(parse_header())
    src = tmp_path / 'host.c'
    src.write_text('#include <stdio.h>\n#include "dbw_hip.h"\n'
                   'int main(void) {\n'
                   'typedef void (*fn_t)(void);\n'
                   '    fn_t fns[] = {\n' + ',\n'.join(f'        (fn_t){n}' for n in names) + '};\n'
                   '    unsigned long k = 0; for (unsigned i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) k += fns[i] != 0;\n'
                   '    printf("%d %lu %s\\n", dbw_abi_version(), k, dbw_last_error() ? "ok" : "null");\n'
                   '    return 0;\n}\n')
    exe = tmp_path / 'host'
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe),
                        '-L', libdir, '-ldbw_hip', f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == _lib.ABI_VERSION and int(out[1]) == len(names) and out[2] == 'ok'
