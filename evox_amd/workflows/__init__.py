from .std_workflow import StdWorkflow, StdWorkflowState
