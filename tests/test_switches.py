"""The run-time switch table (csrc/switches.h): ONE table, filled from the environment once at library load,
changed afterwards only through the C ABI; no getenv / os.environ read on any call path of the product."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'fcd_gan_pytorch_amd')


def test_table_lists_every_switch_with_default_and_help():
    from fcd_gan_pytorch_amd import _lib
    rows = _lib.switch_table()
    names = [r[0] for r in rows]
    assert len(names) == len(set(names)) == _lib.lib.fcd_switch_count() >= 50
    for name, value, default, help_ in rows:
        assert re.fullmatch(r'[A-Z0-9_]+', name) and help_ and value >= 0 and default >= 0
    # the header's X-macro is the only definition: every name in it is in the table, in order
    src = open(os.path.join(PKG, 'csrc', 'switches.h')).read()
    assert re.findall(r'^\s*X\(([A-Z0-9_]+),', src, flags=re.M) == names


def test_set_get_restore_and_the_named_setters_share_the_table():
    from fcd_gan_pytorch_amd import _lib
    lib = _lib.lib
    assert _lib.switch('WGRAD_SPLIT') == _lib.switch('FCD_WGRAD_SPLIT') == lib.fcd_conv_wgrad_split_set(-1)
    with _lib.switched(WGRAD_SPLIT=0, WINO_SPLIT=0, WINO=2, D_POOL=2):
        assert lib.fcd_conv_wgrad_split_set(-1) == 0 and lib.fcd_conv_wino_split_set(-1) == 0 and lib.fcd_conv_wino_set(-1) == 2
        assert _lib.switch('D_POOL') == 2
    assert lib.fcd_conv_wgrad_split_set(-1) == 1 and lib.fcd_conv_wino_split_set(-1) == 1 and lib.fcd_conv_wino_set(-1) == 4
    old = lib.fcd_conv_wgrad_split_set(0)
    assert old == 1 and _lib.switch('WGRAD_SPLIT') == 0
    assert _lib.set_switch('WGRAD_SPLIT', None) == 0 and _lib.switch('WGRAD_SPLIT') == 1        # None / < 0: back to the default
    assert _lib.set_switch('WINO', 3) == 4 and _lib.switch('WINO') == 4                          # range rule: only 0 / 2 / 4
    assert _lib.set_switch('WINO', 4) == 4
    with pytest.raises(_lib.FcdError):
        _lib.switch('NO_SUCH_SWITCH')
    with pytest.raises(_lib.FcdError):
        _lib.set_switch('NO_SUCH_SWITCH', 1)
    assert b'NO_SUCH_SWITCH' in lib.fcd_last_error_string()


def test_environment_is_read_once_at_load():
    code = ("import os; from fcd_gan_pytorch_amd import _lib; a = (_lib.switch('WINO'), _lib.switch('D_POOL'), _lib.switch('WGRAD_WGS'), "
            "_lib.switch('BN_FUSE')); os.environ['FCD_WINO'] = '0'; os.environ['FCD_BN_FUSE'] = '1'; "
            "print(a, (_lib.switch('WINO'), _lib.switch('BN_FUSE')))")
    env = dict(os.environ, FCD_WINO='2', FCD_D_POOL='pooled', FCD_WGRAD_WGS='-7', FCD_BN_FUSE='0', PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    assert out.splitlines()[-1] == '(2, 2, 512, 0) (2, 0)'          # parsed + range-cleaned at load; later setenv changes nothing


def test_no_environment_read_on_any_call_path():
    hits = []
    for f in sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')) + glob.glob(os.path.join(PKG, 'csrc', '*.h'))):
        for i, line in enumerate(open(f), 1):
            if 'getenv' in line and not line.lstrip().startswith('//'):
                hits.append('%s:%d' % (os.path.basename(f), i))
    assert hits == [h for h in hits if h.startswith('common.hip:')] and len(hits) == 1, hits      # the load-time constructor
    allowed = {'FCD_LIB', 'FCDGAN_VGG16_WEIGHTS', 'TORCH_HOME'}
    for f in sorted(glob.glob(os.path.join(PKG, '*.py'))):
        for m in re.finditer(r"environ(?:\.get)?[\[(]\s*'([A-Z0-9_]+)'", open(f).read()):
            assert m.group(1) in allowed or not m.group(1).startswith('FCD'), (os.path.basename(f), m.group(1))


def test_launch_window_logic_on_fake_events(monkeypatch):
    """_ops._LaunchWindow without a device: every 128th launch records an event, the issuing thread polls the OLDEST one (sleeping) while
    more than LAUNCH_WINDOW launches are outstanding, LAUNCH_WINDOW=0 drops everything."""
    import torch
    from fcd_gan_pytorch_amd import _lib, _ops
    log = []

    class FakeEvent:
        def __init__(self):
            self.polls = 0

        def record(self, stream):
            log.append(('record', stream))

        def query(self):
            self.polls += 1
            return self.polls >= 3          # "done" at the third poll

    monkeypatch.setattr(torch.cuda, 'Event', FakeEvent)
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: False)
    sleeps = []
    monkeypatch.setattr(_ops.time, 'sleep', lambda s: sleeps.append(s))
    win = _ops._LaunchWindow()
    with _lib.switched(LAUNCH_WINDOW=256):
        for i in range(128 * 5):
            win.note('s0')
            assert len(win.events) <= 2
        assert len(log) == 5 and len(win.events) == 2 and len(sleeps) == 3 * 2       # three waits, two sleeping polls each
    with _lib.switched(LAUNCH_WINDOW=0):
        for i in range(128):
            win.note('s0')
        assert len(win.events) == 0 and len(log) == 5
