"""Minimal stand-in for `timm` (absent from this image); test infrastructure only."""
