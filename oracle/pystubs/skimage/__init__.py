# empty stand-in so the unmodified pyredner package imports offline (pyredner/image.py:2-6)
