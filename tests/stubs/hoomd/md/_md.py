class CustomForceCompute:
    """What HOOMD's integrator holds for a force.Custom: calls back into Python once per step."""

    def __init__(self, sysdef, callback, aniso):
        self.sysdef, self.callback, self.aniso = sysdef, callback, aniso

    def compute(self, timestep):
        self.callback(timestep)
