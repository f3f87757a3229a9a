"""Minimal stand-in for the `diffusers` package -- TEST TOOLING ONLY.

Purpose: let the *unmodified* reference files under /root/reference import in this container (diffusers is not
installed and there is no network) so that tools/make_goldens.py can record golden vectors from them.  Only the base
classes / helpers the reference touches are provided; no model code lives here.  Never imported by worldforge_amd/.
"""
__version__ = "0.35.1-shim"
