"""hoomd.custom stand-in: the base class user-defined actions derive from.  A container, nothing else."""


class Action:
    def __init__(self):
        self._state = None

    def attach(self, simulation):
        self._state = simulation.state

    def detach(self):
        self._state = None

    def act(self, timestep):
        raise NotImplementedError
