"""CPU: the built C-ABI library loads and exports every symbol include/foundationpose_amd.h declares
(no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def header_symbols():
  src = open(os.path.join(REPO, 'include', 'foundationpose_amd.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(fp_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def built():
  import __graft_entry__ as g
  g.build()
  from foundationpose_amd import _lib
  return _lib


def test_header_and_binding_agree(built):
  assert header_symbols() == built.exported_symbols()


def test_library_exports_every_symbol(built):
  L = ctypes.CDLL(built.LIB_PATH)
  for s in header_symbols():
    assert hasattr(L, s), f'{s} missing from libfoundationpose_amd.so'
  assert built.lib().fp_version() >= 100


def test_only_the_c_abi_is_exported(built):
  """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table holds the C ABI and nothing else (no C++ helper, no kernel
  handle that another HIP library in the host process could interpose)."""
  import subprocess
  out = subprocess.run(['nm', '-D', '--defined-only', built.LIB_PATH], capture_output=True, text=True, check=True).stdout
  names = [l.split()[-1] for l in out.splitlines() if l.strip()]
  assert names and all(n.startswith('fp_') for n in names), [n for n in names if not n.startswith('fp_')][:10]
  assert sorted(n for n in names if not n.startswith('fp_dbg_')) == header_symbols()


def test_product_does_not_import_oracle():
  """The product path must never route through the CPU oracle."""
  pkg = os.path.join(REPO, 'foundationpose_amd')
  for root, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h', '.cpp')):
        txt = open(os.path.join(root, f)).read()
        assert 'import oracle' not in txt and 'from oracle' not in txt, f'{f} references the oracle'


def test_missing_library_fails_loudly(monkeypatch, built):
  monkeypatch.setattr(built, '_lib', None)
  monkeypatch.setattr(built, 'LIB_PATH', '/nonexistent/libfoundationpose_amd.so')
  with pytest.raises(built.FoundationPoseAmdError):
    built.lib()


def test_launch_attributes_are_set_per_device_not_behind_process_wide_flags():
  """hipFuncAttributeMaxDynamicSharedMemorySize belongs to a function on ONE device: it is set for every kernel of the library in
  fp_set_kernel_attributes (called by fp_ctx_create on the context's device), never lazily behind a `static bool` in a launcher."""
  csrc = os.path.join(REPO, 'foundationpose_amd', 'csrc')
  setters = []
  for f in sorted(os.listdir(csrc)):
    if not f.endswith('.hip'):
      continue
    txt = open(os.path.join(csrc, f)).read()
    assert not re.search(r'static\s+bool\s+\w*(attr|set)\w*\s*=', txt), f'{f}: a process-wide flag guards a per-device attribute'
    setters += [f] * len(re.findall(r'hipFuncSetAttribute\s*\(', txt))
  assert setters == ['api.hip'], setters
  api = open(os.path.join(csrc, 'api.hip')).read()
  body = api[api.index('int fp_set_kernel_attributes'):api.index('extern "C" int fp_ctx_create')]
  for fn in re.findall(r'void (\w+_kernel_lds)\(', open(os.path.join(csrc, 'common.h')).read()):
    assert fn + '(v)' in body, f'{fn} is declared in common.h but fp_set_kernel_attributes does not call it'
