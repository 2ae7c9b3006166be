"""CPU oracle for the HistoGAN hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``histogan_amd/``, ``histogram_classes/`` or ``histoGAN/`` may
import this package: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker / the
timed CPU baseline -- never as the thing shipped.
"""
