def log(func=None, *, requires_run=False, category=None):
    """hoomd.logging.log: turns a method into a loggable property."""
    def wrap(f):
        p = property(f)
        return p
    return wrap(func) if func is not None else wrap
