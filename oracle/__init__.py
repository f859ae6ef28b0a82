"""CPU oracle for the DiFashion denoising hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``difashion_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline -- never as the shipped path.

Parity status (see DESIGN.md "Oracle"):
  * glue (MutualEncoder, training loss assembly, CFG sampler, compute_snr):
    PINNED against golden vectors captured from the real
    ``/root/reference/DiFashion/models/difashion.py`` (tests/golden/).
  * U-Net and DDIM/PNDM arithmetic: the reference delegates these to
    third-party ``diffusers==0.18.2`` (pin: reference README.md:27), which is
    neither vendored under /root/reference nor installed.  The restatement
    follows the published diffusers 0.18.2 algorithm and is checked
    structurally (parameter census 859.53 M / 865.92 M, key/shape table) and
    against closed-form known answers.  -> "parity unpinned" for those two.
"""
