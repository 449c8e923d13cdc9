#!/usr/bin/env python3
"""Print the run-time switch table of libfcdgan_hip.so (csrc/switches.h) -- the ONE place the switches are defined.
    python tools/list_switches.py            # markdown table: name, default, current value if different, meaning
README.md's list is this output; tests/test_switches.py checks that the header's X-macro and the loaded table agree."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fcd_gan_pytorch_amd import _lib
    print('| switch (env `FCD_<name>` at load, `fcd_switch_set` afterwards) | default | meaning |')
    print('|---|---|---|')
    for name, value, default, help_ in _lib.switch_table():
        now = '' if value == default else ' (now %d)' % value
        print('| `%s` | %d%s | %s |' % (name, default, now, help_))


if __name__ == '__main__':
    main()
