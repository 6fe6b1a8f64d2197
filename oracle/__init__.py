"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the bar-VAE hot path.

Everything under ``oracle/`` is a checker.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product (``musicgeneration_vae-torch_amd/``) never does: it runs on
the HIP library or raises.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own
``graph/*.py`` from ``/root/reference`` in the build container, loads identical
weights into both, asserts bit-equal outputs, and writes the small fixtures in
``tests/golden/``.  Pieces of the reference that cannot run on CPU (``Loss``,
``model_with_gan.Model.forward``; SURVEY.md defects D2/D3) are restated from the
source text and pinned only by composition of pinned pieces; those are marked
"unpinned" where they are defined.
"""
