class Device:
    pass


class GPU(Device):
    pass


class CPU(Device):
    pass
