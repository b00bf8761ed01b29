from .evoxbench import C10MOP, CitySegMOP, EvoXBenchProblem, IN1kMOP

__all__ = ["C10MOP", "CitySegMOP", "EvoXBenchProblem", "IN1kMOP"]
