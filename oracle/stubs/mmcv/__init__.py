"""Minimal stand-in for `mmcv` (absent from this image); test infrastructure only."""
