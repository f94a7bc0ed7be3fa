"""find_trainer_using_model_name (trainers/__init__.py:4-25 of the reference)."""


def find_trainer_using_model_name(model_name):
    if model_name == "defectgan":
        from .defectgan_trainer import DefectGanTrainer
        return DefectGanTrainer
    if model_name == "mae":
        from .mae_trainer import MAETrainer
        return MAETrainer
    raise ValueError(f"trainer for [{model_name}] is not part of the MI355X hot path ('defectgan' | 'mae')")
