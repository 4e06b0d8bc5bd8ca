# empty stand-in so the unmodified pyredner package imports offline
