"""placeholder — filled in below"""
