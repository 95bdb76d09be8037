"""deep_contact_estimator_amd -- MI355X-native sliding-window contact-state inference.

Drop-in for the reference's inference path (contact_cnn.forward + the windowed batching
loops); see DESIGN.md and INTEGRATION.md.  Importing the package does not need a GPU or the
built library; using the model does (there is no CPU fallback).
"""
from .contact_cnn import contact_cnn, load_checkpoint          # noqa: F401
from . import synth                                              # noqa: F401

__all__ = ["contact_cnn", "load_checkpoint", "synth"]
