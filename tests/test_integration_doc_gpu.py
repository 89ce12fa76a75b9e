"""INTEGRATION.md's ctypes stub is executable documentation: run it as written
(only the library path is made absolute) against the golden rule vectors."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_runs(golden_rules):
    import iago_amd._lib  # noqa: F401  (loads torch's HIP runtime first, as section 3 explains)
    iago_amd._lib.lib()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libiago_hip.so")',
                        'C.CDLL(%r)' % os.path.join(ROOT, "iago_amd", "libiago_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    gf = ns["GameFunctions"]
    for rec in golden_rules["trace"][::211]:
        p1, p2, color, legal, action, q1, q2 = (int(x) for x in rec)
        s = orc.bits_to_state(p1, p2)
        assert orc.actions_to_mask(gf.legal_actions(s, color)) == legal
        a = -1 if action == 0xFF else action
        out = gf.place_stone(s, a, color)
        assert out is s and orc.state_to_bits(s) == (q1, q2)
