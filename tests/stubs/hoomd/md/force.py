"""hoomd.md.force stand-in: Force (attach machinery) and Custom (local force arrays)."""
import contextlib

from . import _md


class Force:
    def __init__(self):
        self._param_dict = {}
        self._simulation = None
        self._cpp_obj = None
        self._attached = False

    def _attach(self, simulation):
        self._simulation = simulation
        self._attach_hook()
        self._attached = True

    def _attach_hook(self):
        pass

    def _detach(self):
        if self._attached:
            self._detach_hook()
        self._attached = False
        self._cpp_obj = None
        self._simulation = None

    def _detach_hook(self):
        pass

    def __getattr__(self, name):
        pd = self.__dict__.get("_param_dict", {})
        if name in pd:
            return pd[name]
        raise AttributeError(name)


class _ForceArrays:
    def __init__(self, force4):
        self.force = force4[:, :3]             # strided views of the Scalar4 buffer, as HOOMD hands them out
        self.potential_energy = force4[:, 3]


class Custom(Force):
    def __init__(self, aniso=False):
        super().__init__()
        self._aniso = aniso

    @property
    def _state(self):
        return self._simulation.state

    def _attach_hook(self):
        self._cpp_obj = _md.CustomForceCompute(self._simulation.state._cpp_sys_def, self.set_forces, self._aniso)

    @property
    @contextlib.contextmanager
    def gpu_local_force_arrays(self):
        if not self._attached or self._cpp_obj is None:
            raise RuntimeError("Cannot access arrays before attaching")   # what the real class does for an unattached force
        yield _ForceArrays(self._simulation.state._force4_for(self))

    @property
    @contextlib.contextmanager
    def cpu_local_force_arrays(self):
        """Host flavour of the same container (numpy Scalar4 buffer); used by tests/golden/make_golden.py when it lets the
        reference's own Python force write into it."""
        if self._cpp_obj is None:
            raise RuntimeError("Cannot access arrays before attaching")
        yield _ForceArrays(self._simulation.state._force4_for(self))
