"""create_model by name (models/__init__.py:6-37 of the reference)."""


def find_model_using_name(model_name):
    if model_name == "defectgan":
        from .defectgan_model import DefectGanModel
        return DefectGanModel
    raise ValueError(f"model [{model_name}] is not part of the MI355X hot path (only 'defectgan')")


def create_model(opt):
    model = find_model_using_name(opt.model)
    instance = model(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
