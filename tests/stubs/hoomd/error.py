"""hoomd.error stand-in: the exception names user code catches."""


class DataAccessError(RuntimeError):
    pass
