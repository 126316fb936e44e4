"""sy11 — the MI355X-native hot path of Spectrogram-YOLOv11 behind the reference's names.

``from sy11 import YOLO`` is the reference's ``from ultralytics import YOLO`` for the detection task."""


def __getattr__(name):                      # lazy: importing the package must not need torch / the HIP library
    if name == "YOLO":
        from .engine.model import YOLO
        return YOLO
    raise AttributeError(name)
