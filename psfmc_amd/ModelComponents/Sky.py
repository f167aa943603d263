from .ComponentBase import ComponentBase, StochasticProperty


class Sky(ComponentBase):
    """Constant sky level in ADU (reference: ModelComponents/Sky.py)."""
    device_kind = 'sky'
    adu = StochasticProperty()

    def __init__(self, adu=None):
        super(Sky, self).__init__()
        self.adu = adu
