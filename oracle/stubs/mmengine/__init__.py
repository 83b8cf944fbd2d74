"""Minimal stand-in for the `mmengine` package (absent from this image).

Test infrastructure only: lets oracle/gen_golden.py import the reference model
package in the build container so golden vectors can be generated from the
reference itself.  Contains no reference code.  Never imported by the product.
"""
from .registry import Registry, MODELS, METRICS  # noqa: F401
