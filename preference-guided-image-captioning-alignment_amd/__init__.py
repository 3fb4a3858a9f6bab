"""MI355X-native hot path of preference-guided image-captioning alignment.

Only what the Stage-1 NT-Xent step and the Stage-2 DPO step need lives here:

* ``csrc/``      hand-written HIP kernels for gfx950 + the C-ABI (``include/pgca_hip.h``)
* ``hip``        ctypes binding of ``libpgca_hip.so`` (fails loudly when the library is missing)
* ``arch``       tower geometries (ViT / GPT-2 families named by the reference configs)
* ``params``     flat HBM parameter store with the reference's ``state_dict`` key names
* ``engine``     forward/backward schedules of the towers on the HIP kernels
* ``model``      the reference's Python surface (``PreferenceGuidedCaptioningModel`` ...)
* ``losses``     ``ContrastiveLoss`` / ``PreferenceLoss`` / ``DPOPreferenceLoss``
* ``trainer``    step semantics (AdamW, cosine warm-up, clip, NaN skip, accumulation)
* ``dist``       RCCL data-parallel engine (bucketed gradient all-reduce, embedding all-gather)

Import as ``pgca_amd`` (see ``pgca_amd/__init__.py``).
"""

__version__ = "0.1.0"
