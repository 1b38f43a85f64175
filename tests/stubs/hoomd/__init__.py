"""STAND-IN for the HOOMD-blue Python package -- NOT HOOMD-blue, and no evidence of API compatibility with it.

The build/test image has no HOOMD-blue.  This package exists for one purpose: to let tests/test_hoomd_plugin_stub.py EXECUTE
the control flow of cav-hoomd_amd/cavitymd/hoomd_plugin.py (attach ladder, Custom route, energy cache, loggables) so that a
typo or a wrong attribute there fails a test instead of waiting for the first user with a real HOOMD-blue.  It mimics only
the handful of names that module touches, with the semantics the reference's own code relies on
(src/cavitymd/forces.py:45-173, src/cavitymd/cavity_force_python.py:31-149).  It is importable only when tests/stubs is put
on sys.path, which only that test does, in a subprocess -- and tests/golden/make_golden.py, in the build container, so that the
REFERENCE's own Python modules (which do `import hoomd` at the top) load and run their own arithmetic on numpy arrays.  For
that second use the rule is strict: this package holds CONTAINERS ONLY (base classes, context managers, a decorator) and not
one line of physics, so every number in tests/golden/reference_python_golden.npz was computed by the reference's bytes.
"""
from . import custom, data, device, error, logging, md  # noqa: F401

IS_STUB = True
