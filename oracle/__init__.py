"""CPU oracle for the VirConv sparse-convolution hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``virconv_amd/``) imports this directory.  The only legal importers
are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- and there
only as the *checker* / reported baseline, never as the thing measured or shipped.

PARITY UNPINNED.  The arithmetic of this path lives in the third-party package ``spconv``
(reference setup.py:41, README.md:52,60,70: spconv-cu111 2.1.22 / 1.2.1, +cumm), which is NOT vendored
under /root/reference, not installed here, and there is no network.  The reference itself ships no
tests, golden vectors or fixtures for this path.  What pins this oracle instead:

  1. ``dense_ref`` -- an independent dense restatement (torch.nn.functional.conv3d/conv2d on the
     densified input, read back at the active sites).  It shares no code with the sparse restatement
     and is the anchor for every sparse operator in ``sparse_ref``.
  2. the reference's OWN composition layer (pcdet/models/backbones_3d/spconv_backbone.py, unmodified,
     imported from /root/reference in the build container) executed on top of these operators to
     produce the committed fixtures in tests/golden/ (generator: tests/golden/make_golden.py).
  3. hand-computable known-answer cases in tests/golden/kat_*.npz.

Modules
  sparse_ref   rulebooks (SubM / strided), gather-GEMM-scatter conv fwd/bwd      
  dense_ref    dense conv oracle O1 + autograd oracle O3
  geometry     voxelizer, MeanVFE, input point discard, layer discard, index2uv  
  backend      adapter exposing the oracle through the product's backend interface (tests only)
"""
