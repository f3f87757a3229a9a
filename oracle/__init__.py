"""CPU oracle for the WorldForge guided-denoising hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Every module here is a plain torch-CPU / numpy restatement of one piece of the reference algorithm
(/root/reference/wan_for_worldforge), written from the reference's arithmetic and citing the file:line it follows.
The oracle is pinned against golden vectors recorded from the *unmodified* reference (tools/make_goldens.py,
fixtures under tests/golden/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
worldforge_amd/ (the product) never does and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  sched.py, inject.py, sampler.py, harness.py : pinned by goldens recorded from the imported reference scheduler/pipeline.
  dit.py  : pinned against the in-tree twin wan/modules/model.py (the executed class is diffusers' WanTransformer3DModel,
            which is not in /root/reference -> "parity unpinned" at the diffusers boundary).
  vae.py  : pinned against the in-tree twin wan/modules/vae.py (executed class: diffusers' AutoencoderKLWan, same caveat).
  Farneback optical flow (cv2) is absent from this container and from /root/reference: parity unpinned; the reference's own
  fallback branch (temporal difference, SCHED:390-392) is what is pinned.
"""
