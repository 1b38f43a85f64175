class ParameterDict(dict):
    def __init__(self, **spec):
        super().__init__()
        self._spec = spec
