"""myolo -- MI355X-native Mask-YOLO hot path behind the reference's Python surface
(``myolo.model.MaskYOLO`` build()/train()/detect(), ``myolo.config.Config``)."""
__version__ = "0.1.0"
