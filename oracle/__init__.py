"""CPU oracle for the DSMnet stereo cost-volume path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dsmnet_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker.

Every function restates one piece of the reference (sunshinnnn/DSMnet) and
cites the reference ``file:line`` it follows.  The restatement is pinned two
ways (see ``tests/golden/make_goldens.py``):

* against the reference's own source executed by this container's torch
  (loaded as text with the shims of SURVEY.md section 8c), and
* through the golden fixtures that script wrote to ``tests/golden/*.npz``.
"""
from . import ops, models  # noqa: F401
