"""Stand-in for OpenCV: the reference imports cv2 at litegs/data.py:6 but only touches it in VideoFrame (video input)."""
CAP_PROP_FPS, CAP_PROP_FRAME_COUNT, CAP_PROP_POS_FRAMES = 5, 7, 1


class VideoCapture:
    def __init__(self, *args, **kwargs):
        raise RuntimeError("compat/cv2.py: video input needs the real OpenCV package")
