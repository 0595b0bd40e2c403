"""CPU oracle for the IODINE refinement hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``iodine_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  The reference ships no tests or golden vectors for
this path, so the restatement is pinned against outputs of
the reference itself, generated in the build container by
``tests/golden/gen_goldens.py`` (imports ``/root/reference/lib/modeling/iodine.py``
unmodified, replays a stored epsilon stream through ``torch.randn_like``) and
committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks
the restatement against every one of those fixtures, and against the single
known-answer vector the reference holds (ARI table -> 0.08333, ``lib/utils/ari.py:56-63``).
"""
