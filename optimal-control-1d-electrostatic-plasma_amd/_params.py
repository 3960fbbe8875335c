"""`update_params(**kwargs)` as the reference's small classes all offer it: set the attributes that
already exist and were given a non-None value, ignore the rest."""


class ParamMixin:
    def update_params(self, **kwargs):
        for name, value in kwargs.items():
            if value is not None and hasattr(self, name):
                setattr(self, name, value)
